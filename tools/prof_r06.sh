#!/bin/bash
# Round 6's profiles: the bench command (kernel stats + FETCH / WRITE passes), every configuration's 1 GiB passes (tools/profile_round.sh), the
# general families at 8 GiB (tools/prof_8g.sh), the one-walk kernel's SQ / FETCH / WRITE passes (tools/prof_one.sh).
#   tools/prof_r06.sh r06   -> gpurun_out/profiles/r06_*
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
bash tools/profile_round.sh $tag > /dev/null 2>&1
bash tools/prof_8g.sh $tag > /dev/null 2>&1
bash tools/prof_one.sh $tag > /dev/null 2>&1
ls gpurun_out/profiles | grep "^$tag" | wc -l
