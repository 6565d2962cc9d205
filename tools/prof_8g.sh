#!/bin/bash
# kernel-trace stats of the general families at the bench's size (8 GiB): tools/prof_8g.sh r04 -> gpurun_out/profiles/<tag>_8g_*.txt
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
out=gpurun_out/profiles
mkdir -p $out gpurun_out/raw
while IFS='|' read -r name kargs; do
    rm -rf gpurun_out/raw/g8_$name
    timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/g8_$name -o s -- python tools/kbench.py $kargs --bytes 8589934592 --steps 5 > gpurun_out/raw/g8_$name.log 2>&1
    { echo "# kbench $kargs --bytes 8589934592 --steps 5"; python tools/rocpd_summary.py gpurun_out/raw/g8_$name/s_results.db trre; grep '^pattern' gpurun_out/raw/g8_$name.log; } > $out/${tag}_8g_${name}_kernel_stats.txt
done <<'CASES'
dict1000_dft|--dict 1000 --engine dft
nft_loop_guided|--case (a|b)*c:x;;nft;;printable;;auto
expand_dft|--case a:xyz;;dft;;printable;;auto
cfg4_nft|--case (cat:dog|dog:cat);;nft;;catdog;;auto
CASES
rm -rf gpurun_out/raw
cat $out/${tag}_8g_*_kernel_stats.txt | cut -c1-70,90-130
