#!/bin/bash
# the round's new kernels under rocprofv3 (kernel stats), cases passed as arrays (patterns with blanks)
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
out=gpurun_out/profiles
mkdir -p $out gpurun_out/raw
run() {
    local name=$1; shift
    rm -rf gpurun_out/raw/n_$name
    timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/n_$name -o s -- python tools/kbench.py "$@" --steps 5 > gpurun_out/raw/n_$name.log 2>&1
    { echo "# kbench $* --steps 5"; python tools/rocpd_summary.py gpurun_out/raw/n_$name/s_results.db trre 2>&1 | tail -n 20; grep '^pattern' gpurun_out/raw/n_$name.log; tail -n 3 gpurun_out/raw/n_$name.log | grep -i "error\|Traceback"; } > $out/${tag}_${name}_kernel_stats.txt
}
run longlines_greedy --case ' +: ;;nft;;long400000;;auto'
run longlines_loop_guided --case '(a|b)*c:x;;nft;;long400000;;auto'
run longlines_cfg4 --case '(cat:dog|dog:cat);;nft;;long400000;;auto'
run dft_lazy --case '(a|b)*a(a|b){18}:x;;dft;;printable;;auto'
run dft_lazy_runs --case '((a:x)*b)|((a:y)*c);;dft;;printable;;auto'
rm -rf gpurun_out/raw
cat $out/${tag}_longlines_*_kernel_stats.txt $out/${tag}_dft_lazy*_kernel_stats.txt | cut -c1-86,96-150
