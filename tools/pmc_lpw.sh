#!/bin/bash
# PMC passes over the window kernel (tools/kbench.py --kernel stream_lp); summaries -> gpurun_out/pmc/<tag>_*.txt
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/pmc
i=0
for set in "${@:2}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d gpurun_out/raw/$i -o p -- python tools/kbench.py --kernel stream_lp --steps 2 > gpurun_out/raw_$i.log 2>&1
  python tools/rocpd_summary.py gpurun_out/raw/$i/p_results.db k_stream_lpw > gpurun_out/pmc/${tag}_$i.txt
done
rm -rf gpurun_out/raw gpurun_out/raw_*.log
cat gpurun_out/pmc/${tag}_*.txt | grep -v "^kernel\|^$"
