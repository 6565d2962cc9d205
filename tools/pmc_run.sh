#!/bin/bash
# PMC passes over a tools/kbench.py run:  tools/pmc_run.sh <tag> <kernel-name filter> "<kbench args>" "<counters>" ["<counters>" ...]
# summaries -> gpurun_out/pmc/<tag>_N.txt
tag=$1; filt=$2; kargs=$3; shift 3
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/pmc
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d gpurun_out/raw/$i -o p -- python tools/kbench.py $kargs --steps 2 > gpurun_out/raw_$i.log 2>&1
  python tools/rocpd_summary.py gpurun_out/raw/$i/p_results.db $filt > gpurun_out/pmc/${tag}_$i.txt
done
rm -rf gpurun_out/raw gpurun_out/raw_*.log
cat gpurun_out/pmc/${tag}_*.txt | grep -v "^kernel\|^$"
