#!/bin/bash
# SQ / FETCH / WRITE passes and kernel stats of the memoryless one-pass kernel (round 6, TRRE_MAPGEN=1), 1 GiB: 'a:xyz' and '[aie]:'
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
out=gpurun_out/profiles
export TRRE_MAPGEN=1
run() {   # name, suffix, rocprof args..., -- case
    local name=$1 suf=$2; shift 2
    local args=()
    while [ "$1" != "--" ]; do args+=("$1"); shift; done
    shift
    mkdir -p $out gpurun_out/raw
    rm -rf gpurun_out/raw/q_$name
    timeout 300 rocprofv3 "${args[@]}" -d gpurun_out/raw/q_$name -o p -- python tools/kbench.py --case "$1" --steps ${STEPS:-2} > gpurun_out/raw/q_$name.log 2>&1
    { echo "# TRRE_MAPGEN=1 kbench --case '$1' --steps ${STEPS:-2}   (rocprofv3 ${args[*]})"; python tools/rocpd_summary.py gpurun_out/raw/q_$name/p_results.db trre 2>&1 | tail -n 40; grep '^pattern' gpurun_out/raw/q_$name.log; } > $out/${tag}_${name}_$suf.txt
}
for c in "expand_map|a:xyz;;dft;;printable;;auto" "delete_map|[aie]:;;nft;;printable;;auto"; do
    n=${c%%|*}; k=${c#*|}
    STEPS=5 run $n kernel_stats --kernel-trace --stats -- "$k"
    run $n pmc_FETCH_SIZE --pmc FETCH_SIZE -- "$k"
    run $n pmc_WRITE_SIZE --pmc WRITE_SIZE -- "$k"
    run $n pmc_sq_1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES -- "$k"
    run $n pmc_sq_2 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -- "$k"
    run $n pmc_sq_3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -- "$k"
done
rm -rf gpurun_out/raw
for f in $out/${tag}_expand_map_* $out/${tag}_delete_map_kernel*; do echo "== $f"; cut -c1-86,96-150 $f | grep -v "^$" | head -30; done
