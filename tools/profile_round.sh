#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02
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
    timeout 400 rocprofv3 "$@" > gpurun_out/raw/$name.log 2>&1
}
# 1. kernel-trace stats of the bench command itself (headline workload, 8 GiB per GPU)
run stats --kernel-trace --stats -d gpurun_out/raw/stats -o s -- python bench.py --steps 50 --warmup 2 --no-cpu --no-extras
python tools/rocpd_summary.py gpurun_out/raw/stats/s_results.db trre > $out/${tag}_bench_kernel_stats.txt
grep '^{' gpurun_out/raw/stats.log | tail -n 1 > $out/${tag}_bench_line_under_rocprof.json
# 2. HBM traffic of the dominant kernel (separate PMC passes, no tracing)
for c in FETCH_SIZE WRITE_SIZE; do
    run pmc_$c --pmc $c -d gpurun_out/raw/pmc_$c -o p -- python bench.py --steps 4 --warmup 1 --no-cpu --no-extras
    python tools/rocpd_summary.py gpurun_out/raw/pmc_$c/p_results.db trre > $out/${tag}_bench_pmc_$c.txt
done
# 3. the other configurations on their own corpora, 1 GiB each: kernel-trace stats, then FETCH / WRITE passes
while IFS='|' read -r name kargs; do
    run st_$name --kernel-trace --stats -d gpurun_out/raw/st_$name -o s -- python tools/kbench.py $kargs --steps 5
    { echo "# kbench $kargs --steps 5"; python tools/rocpd_summary.py gpurun_out/raw/st_$name/s_results.db trre; grep '^pattern' gpurun_out/raw/st_$name.log; } > $out/${tag}_${name}_kernel_stats.txt
    for c in FETCH_SIZE WRITE_SIZE; do
        run pm_${name}_$c --pmc $c -d gpurun_out/raw/pm_${name}_$c -o p -- python tools/kbench.py $kargs --steps 2
        { echo "# kbench $kargs --steps 2   (rocprofv3 --pmc $c)"; python tools/rocpd_summary.py gpurun_out/raw/pm_${name}_$c/p_results.db trre; } > $out/${tag}_${name}_pmc_$c.txt
    done
done <<'CASES'
cfg4_nft|--case (cat:dog|dog:cat);;nft;;catdog;;auto
cfg4_nft_guided|--case (cat:dog|dog:cat);;nft;;catdog;;guided_lp
expand_dft|--case a:xyz;;dft;;printable;;auto
nft_loop_guided|--case (a|b)*c:x;;nft;;printable;;auto
dft_loop_guided|--case (a|b)*c:x;;dft;;printable;;auto
tile_dft|--case a:xyz;;dft;;printable;;tile_gen --bytes 268435456
wide_guided|--case a(a|b|c|d|e|f|g|h){9}c:x;;nft;;printable;;auto --bytes 268435456
backtrack|--case a(a|b|c|d|e|f|g|h){12}c:x;;nft;;printable;;auto --bytes 268435456
CASES
# 5. the dictionary configuration: kernel stats with the splice and with the older copy pass, SQ counters and traffic (tools/prof_dict4.sh, tools/pmc_dict4.sh)
bash tools/prof_dict4.sh $tag > /dev/null 2>&1
bash tools/pmc_dict4.sh $tag > /dev/null 2>&1
rm -rf gpurun_out/raw
ls -la $out
