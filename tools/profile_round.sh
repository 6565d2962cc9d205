#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# Writes text summaries (kernel-trace stats + PMC traffic passes) to gpurun_out/profiles/<tag>_*.txt;
# copy the ones to be judged into profiles/.
set -u
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
out=gpurun_out/profiles
mkdir -p $out gpurun_out/raw
run() {  # name, pmc-or-stats args..., -- command
    local name=$1; shift
    rm -rf gpurun_out/raw/$name
    timeout 300 rocprofv3 "$@" > gpurun_out/raw/$name.log 2>&1
}
# 1. kernel-trace stats of the bench command itself
run stats --kernel-trace --stats -d gpurun_out/raw/stats -o s -- python bench.py --steps 10 --warmup 2 --no-cpu --no-extras
python tools/rocpd_summary.py gpurun_out/raw/stats/s_results.db trre > $out/${tag}_bench_kernel_stats.txt
grep '^{' gpurun_out/raw/stats.log | tail -n 1 > $out/${tag}_bench_line_under_rocprof.json
# 2. HBM traffic of the dominant kernel (separate PMC passes, no tracing)
for c in FETCH_SIZE WRITE_SIZE; do
    run pmc_$c --pmc $c -d gpurun_out/raw/pmc_$c -o p -- python bench.py --steps 4 --warmup 1 --no-cpu --no-extras
    python tools/rocpd_summary.py gpurun_out/raw/pmc_$c/p_results.db trre > $out/${tag}_bench_pmc_$c.txt
done
# 3. the stream kernels: the cfg4 pattern (NFT, window kernel), an expanding pattern and the cfg5-style
#    dictionary (count + emit passes)
for spec in "cfg4_nft|--pattern (cat:dog|dog:cat) --engine nft" "expand_dft|--pattern a:xyz --engine dft" "dict1000_dft|--dict 1000 --engine dft"; do
    name=${spec%%|*}; kargs=${spec#*|}
    run st_$name --kernel-trace --stats -d gpurun_out/raw/st_$name -o s -- python tools/kbench.py $kargs --steps 5
    python tools/rocpd_summary.py gpurun_out/raw/st_$name/s_results.db trre > $out/${tag}_${name}_kernel_stats.txt
    grep '^pattern' gpurun_out/raw/st_$name.log >> $out/${tag}_${name}_kernel_stats.txt
done
# 4. HBM traffic of the window kernel on the cfg4 pattern
for c in FETCH_SIZE WRITE_SIZE; do
    run pmcw_$c --pmc $c -d gpurun_out/raw/pmcw_$c -o p -- python tools/kbench.py --pattern "(cat:dog|dog:cat)" --engine nft --steps 3
    python tools/rocpd_summary.py gpurun_out/raw/pmcw_$c/p_results.db k_stream_lpw > $out/${tag}_cfg4_nft_pmc_$c.txt
done
rm -rf gpurun_out/raw
ls -la $out
