#!/bin/bash
# Round 5's profiles: the bench command (kernel stats + FETCH / WRITE passes), the general families at 8 GiB, and the round's new
# kernels — exact sub-ranges on long lines, the lazy family, the backward pass's verify — at 1 GiB, with FETCH / WRITE / SQ passes for the
# exact sub-range kernels.   tools/prof_r05.sh r05   -> gpurun_out/profiles/r05_*
set -u
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
out=gpurun_out/profiles
mkdir -p $out gpurun_out/raw
bash tools/prof_bench.sh $tag > /dev/null 2>&1
bash tools/prof_8g.sh $tag > /dev/null 2>&1
while IFS='|' read -r name kargs; do
    rm -rf gpurun_out/raw/n_$name
    timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/n_$name -o s -- python tools/kbench.py $kargs --steps 5 > gpurun_out/raw/n_$name.log 2>&1
    { echo "# kbench $kargs --steps 5"; python tools/rocpd_summary.py gpurun_out/raw/n_$name/s_results.db trre; grep '^pattern' gpurun_out/raw/n_$name.log; } > $out/${tag}_${name}_kernel_stats.txt
done <<'CASES'
longlines_greedy|--case  +: ;;nft;;long400000;;auto
longlines_loop_guided|--case (a|b)*c:x;;nft;;long400000;;auto
longlines_cfg4|--case (cat:dog|dog:cat);;nft;;long400000;;auto
dft_lazy|--case (a|b)*a(a|b){18}:x;;dft;;printable;;auto
dft_lazy_runs|--case ((a:x)*b)|((a:y)*c);;dft;;printable;;auto
expand_dft|--case a:xyz;;dft;;printable;;auto
nft_loop_guided|--case (a|b)*c:x;;nft;;printable;;auto
CASES
i=0
for set in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    for name in "expand_dft|--case a:xyz;;dft;;printable;;auto" "nft_loop_guided|--case (a|b)*c:x;;nft;;printable;;auto" "longlines_loop_guided|--case (a|b)*c:x;;nft;;long400000;;auto"; do
        n=${name%%|*}; kargs=${name#*|}
        rm -rf gpurun_out/raw/p_$n
        timeout 300 rocprofv3 --pmc $set -d gpurun_out/raw/p_$n -o p -- python tools/kbench.py $kargs --steps 2 > gpurun_out/raw/p_$n.log 2>&1
        c=$(echo $set | cut -d' ' -f1); [ $i = 3 ] && c=sq
        { echo "# kbench $kargs --steps 2   (rocprofv3 --pmc $set)"; python tools/rocpd_summary.py gpurun_out/raw/p_$n/p_results.db trre; } > $out/${tag}_${n}_pmc_$c.txt
    done
done
rm -rf gpurun_out/raw
ls $out | grep "^$tag" | wc -l
