bash tools/prof_dict4.sh r04d 2>&1 | grep "k_fb\|GB/s" | cut -c1-160 | head -2
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
TRRE_EMIT_DBG=8 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/x -o s -- python tools/kbench.py --dict 1000 --engine dft --steps 5 > gpurun_out/raw/x.log 2>&1
python tools/rocpd_summary.py gpurun_out/raw/x/s_results.db trre | cut -c1-150 | grep "k_fb"
rm -rf gpurun_out/raw
