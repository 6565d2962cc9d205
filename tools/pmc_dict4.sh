# round 4: SQ counters of the dictionary configuration's kernels (separate --pmc passes, no tracing)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/profiles; mkdir -p $out gpurun_out/raw
tag=${1:-r04}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d gpurun_out/raw/sq_$i -o p -- python tools/kbench.py --dict 1000 --engine dft --steps 2 > gpurun_out/raw/sq_$i.log 2>&1
  { echo "# kbench --dict 1000 --engine dft --steps 2   (rocprofv3 --pmc $set)"; python tools/rocpd_summary.py gpurun_out/raw/sq_$i/p_results.db fb_; } > $out/${tag}_dict1000_dft_pmc_sq_$i.txt
  cat $out/${tag}_dict1000_dft_pmc_sq_$i.txt | cut -c1-200
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d gpurun_out/raw/pm_$c -o p -- python tools/kbench.py --dict 1000 --engine dft --steps 2 > gpurun_out/raw/pm_$c.log 2>&1
  { echo "# kbench --dict 1000 --engine dft --steps 2   (rocprofv3 --pmc $c)"; python tools/rocpd_summary.py gpurun_out/raw/pm_$c/p_results.db trre; } > $out/${tag}_dict1000_dft_pmc_$c.txt
  cat $out/${tag}_dict1000_dft_pmc_$c.txt | cut -c1-200
done
rm -rf gpurun_out/raw
