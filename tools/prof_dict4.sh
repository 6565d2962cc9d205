# kernel stats of the dictionary configuration (mark + splice; round 4 also ran round 3's copy pass beside it: removed in round 6)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/profiles; mkdir -p $out gpurun_out/raw
tag=${1:-r04}
run() { # name, env
  env $2 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/st_$1 -o s -- python tools/kbench.py --dict 1000 --engine dft --steps 5 > gpurun_out/raw/st_$1.log 2>&1
  { echo "# $2 kbench --dict 1000 --engine dft --steps 5"; python tools/rocpd_summary.py gpurun_out/raw/st_$1/s_results.db trre; grep '^pattern' gpurun_out/raw/st_$1.log; } > $out/${tag}_dict1000_dft_$1_kernel_stats.txt
  cat $out/${tag}_dict1000_dft_$1_kernel_stats.txt | cut -c1-150
}
run splice A=1
rm -rf gpurun_out/raw
