#!/bin/bash
# Round 5's profiles: the bench command (kernel stats + FETCH / WRITE passes), the general families at 8 GiB, the round's new kernels —
# exact sub-ranges on long lines, the lazy family — at 1 GiB (prof_r05b.sh), the FETCH / WRITE / SQ passes of the exact sub-range
# kernels (prof_r05c.sh) and the counter calibration (calib_run.sh).   tools/prof_r05.sh r05   -> gpurun_out/profiles/r05_*
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
bash tools/prof_bench.sh $tag > /dev/null 2>&1
bash tools/prof_8g.sh $tag > /dev/null 2>&1
bash tools/prof_r05b.sh $tag > /dev/null 2>&1
bash tools/prof_r05c.sh $tag > /dev/null 2>&1
bash tools/calib_run.sh $tag > /dev/null 2>&1
ls gpurun_out/profiles | grep "^$tag" | wc -l
