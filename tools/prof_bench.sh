#!/bin/bash
# The headline part of tools/profile_round.sh on its own: kernel-trace stats and the two traffic passes of the bench command.
#   tools/prof_bench.sh r03      (through gpurun, from the repo root) -> gpurun_out/profiles/<tag>_bench_*
set -u
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/profiles
mkdir -p $out gpurun_out/raw
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/stats -o s -- python bench.py --steps 50 --warmup 2 --no-cpu --no-extras > gpurun_out/raw/stats.log 2>&1
python tools/rocpd_summary.py gpurun_out/raw/stats/s_results.db trre > $out/${tag}_bench_kernel_stats.txt
grep '^{' gpurun_out/raw/stats.log | tail -n 1 > $out/${tag}_bench_line_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c -d gpurun_out/raw/pmc_$c -o p -- python bench.py --steps 4 --warmup 1 --no-cpu --no-extras > gpurun_out/raw/pmc_$c.log 2>&1
    python tools/rocpd_summary.py gpurun_out/raw/pmc_$c/p_results.db trre > $out/${tag}_bench_pmc_$c.txt
done
rm -rf gpurun_out/raw
cat $out/${tag}_bench_kernel_stats.txt $out/${tag}_bench_pmc_FETCH_SIZE.txt $out/${tag}_bench_pmc_WRITE_SIZE.txt | cut -c1-170
