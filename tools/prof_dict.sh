cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/profiles; mkdir -p $out gpurun_out/raw
for eng in dft nft; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/st_$eng -o s -- python tools/kbench.py --dict 1000 --engine $eng --steps 5 > gpurun_out/raw/st_$eng.log 2>&1
  { echo "# kbench --dict 1000 --engine $eng --steps 5"; python tools/rocpd_summary.py gpurun_out/raw/st_$eng/s_results.db trre; grep '^pattern' gpurun_out/raw/st_$eng.log; } > $out/r03_dict1000_${eng}_kernel_stats.txt
done
TRRE_NO_FB_COPY=1 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/st_old -o s -- python tools/kbench.py --dict 1000 --engine dft --steps 5 > gpurun_out/raw/st_old.log 2>&1
{ echo "# TRRE_NO_FB_COPY=1 kbench --dict 1000 --engine dft --steps 5   (the count / emit pair: what the copy form replaced, and its fallback)"; python tools/rocpd_summary.py gpurun_out/raw/st_old/s_results.db trre; grep '^pattern' gpurun_out/raw/st_old.log; } > $out/r03_dict1000_dft_nocopy_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d gpurun_out/raw/pm_$c -o p -- python tools/kbench.py --dict 1000 --engine dft --steps 2 > gpurun_out/raw/pm_$c.log 2>&1
  { echo "# kbench --dict 1000 --engine dft --steps 2   (rocprofv3 --pmc $c)"; python tools/rocpd_summary.py gpurun_out/raw/pm_$c/p_results.db trre; } > $out/r03_dict1000_dft_pmc_$c.txt
done
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d gpurun_out/raw/sq_$i -o p -- python tools/kbench.py --dict 1000 --engine dft --steps 2 > gpurun_out/raw/sq_$i.log 2>&1
  { echo "# kbench --dict 1000 --engine dft --steps 2   (rocprofv3 --pmc $set)"; python tools/rocpd_summary.py gpurun_out/raw/sq_$i/p_results.db fb_; } > $out/r03_dict1000_dft_pmc_sq_$i.txt
done
rm -rf gpurun_out/raw
cat $out/r03_dict1000_dft_kernel_stats.txt $out/r03_dict1000_dft_pmc_FETCH_SIZE.txt $out/r03_dict1000_dft_pmc_WRITE_SIZE.txt | cut -c1-160
