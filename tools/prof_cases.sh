#!/bin/bash
# kernel-trace stats of tools/kbench.py cases, one rocprofv3 run per case:
#   tools/prof_cases.sh <tag> "<kbench args>" ["<kbench args>" ...]
# summaries -> gpurun_out/prof/<tag>_N.txt (and stdout)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/prof gpurun_out/raw
i=0
for kargs in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/raw/$i
  eval timeout 250 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/$i -o s -- python tools/kbench.py $kargs > gpurun_out/raw/$i.log 2>&1
  { echo "# kbench $kargs"; python tools/rocpd_summary.py gpurun_out/raw/$i/s_results.db trre; grep '^pattern' gpurun_out/raw/$i.log; } > gpurun_out/prof/${tag}_$i.txt
  cat gpurun_out/prof/${tag}_$i.txt
done
rm -rf gpurun_out/raw
