cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/raw
for v in A TRRE_NO_G16_SPLICE; do
for c in "a:xyz;;dft;;printable;;auto" "(a|b)*c:x;;nft;;printable;;auto"; do
env $v=1 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/x -o s -- python tools/kbench.py --case "$c" --steps 5 > gpurun_out/raw/x.log 2>&1
echo "== $v $c"; python tools/rocpd_summary.py gpurun_out/raw/x/s_results.db trre | cut -c1-150 | grep -v "^kernel\|^$"
rm -rf gpurun_out/raw/x
done; done
rm -rf gpurun_out/raw
