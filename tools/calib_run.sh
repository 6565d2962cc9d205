#!/bin/bash
# PMC calibration (tools/probes/fetch_calib.hip): FETCH_SIZE / WRITE_SIZE per access pattern over a known byte count.
#   tools/calib_run.sh r05   ->  gpurun_out/profiles/r05_fetch_calibration.txt
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
mkdir -p gpurun_out/profiles gpurun_out/raw
out=gpurun_out/profiles/${tag}_fetch_calibration.txt
{
echo "# tools/probes/fetch_calib under rocprofv3 --pmc: every kernel moves 1 GiB = 1048576 KB in one access pattern (lanes 4 KiB apart)."
echo "# FETCH_SIZE / WRITE_SIZE are in KB; factor = 1048576 / counter = what bench.py multiplies the counter by for kernels that move their bytes that way."
} > $out
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    name=$(echo $c | tr ' ' '_')
    rm -rf gpurun_out/raw/cal_$name
    timeout 200 rocprofv3 --pmc $c -d gpurun_out/raw/cal_$name -o p -- tools/probes/fetch_calib > gpurun_out/raw/cal_$name.log 2>&1
    { echo; echo "## rocprofv3 --pmc $c"; python tools/rocpd_summary.py gpurun_out/raw/cal_$name/p_results.db cal_ 2>&1 | grep -v "^kernel" ; } >> $out
done
rm -rf gpurun_out/raw
cat $out
