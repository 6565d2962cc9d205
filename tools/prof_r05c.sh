#!/bin/bash
# FETCH / WRITE / SQ passes and 1 GiB kernel stats for the small-table general families with exact sub-ranges (ordinary text and long lines)
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
out=gpurun_out/profiles
run() {   # name, suffix, rocprof args..., -- case
    local name=$1 suf=$2; shift 2
    local args=()
    while [ "$1" != "--" ]; do args+=("$1"); shift; done
    shift
    mkdir -p $out gpurun_out/raw
    rm -rf gpurun_out/raw/q_$name
    timeout 300 rocprofv3 "${args[@]}" -d gpurun_out/raw/q_$name -o p -- python tools/kbench.py --case "$1" --steps ${STEPS:-2} > gpurun_out/raw/q_$name.log 2>&1
    { echo "# kbench --case '$1' --steps ${STEPS:-2}   (rocprofv3 ${args[*]})"; python tools/rocpd_summary.py gpurun_out/raw/q_$name/p_results.db trre 2>&1 | tail -n 40; grep '^pattern' gpurun_out/raw/q_$name.log; } > $out/${tag}_${name}_$suf.txt
}
for c in "expand_dft|a:xyz;;dft;;printable;;auto" "nft_loop_guided|(a|b)*c:x;;nft;;printable;;auto" "longlines_loop_guided|(a|b)*c:x;;nft;;long400000;;auto"; do
    n=${c%%|*}; k=${c#*|}
    STEPS=5 run $n kernel_stats --kernel-trace --stats -- "$k"
    run $n pmc_FETCH_SIZE --pmc FETCH_SIZE -- "$k"
    run $n pmc_WRITE_SIZE --pmc WRITE_SIZE -- "$k"
    run $n pmc_sq --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR -- "$k"
done
rm -rf gpurun_out/raw
for f in $out/${tag}_expand_dft_* $out/${tag}_nft_loop_guided_* $out/${tag}_longlines_loop_guided_pmc*; do echo "== $f"; cut -c1-86,96-150 $f | head -24; done
