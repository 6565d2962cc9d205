#!/bin/bash
# kernel-trace stats of the general (count + emit) path: the dictionary config and an expanding pattern
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/raw
timeout 250 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/d -o s -- python tools/kbench.py --dict 1000 --engine dft --steps 3 > gpurun_out/raw/d.log 2>&1
python tools/rocpd_summary.py gpurun_out/raw/d/s_results.db trre; grep '^pattern' gpurun_out/raw/d.log
timeout 250 rocprofv3 --kernel-trace --stats -d gpurun_out/raw/x -o s -- python tools/kbench.py --pattern "a:xyz" --steps 3 > gpurun_out/raw/x.log 2>&1
python tools/rocpd_summary.py gpurun_out/raw/x/s_results.db trre; grep '^pattern' gpurun_out/raw/x.log
rm -rf gpurun_out/raw
