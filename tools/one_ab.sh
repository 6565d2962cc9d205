#!/bin/bash
# A/B of the one-pass form (one_block.hpp) against the count / emit pair: same process settings, separate processes (the switches are read once).
#   tools/one_ab.sh [BYTES] [STEPS]      -> gpurun_out/one_ab.log
B=${1:-8589934592}; K=${2:-10}
mkdir -p gpurun_out
{
for one in 0 1; do
  echo "== TRRE_ONE=$one  ${TRRE_ONE_LANE:+lane $TRRE_ONE_LANE} ${TRRE_ONE_REGION:+region $TRRE_ONE_REGION} ${TRRE_ONE_LOOK:+look $TRRE_ONE_LOOK}"
  TRRE_ONE=$one TRRE_TRACE=1 python tools/kbench.py --bytes $B --steps $K --sum \
     --case 'a:xyz;;dft;;printable;;auto' --case ' +: ;;nft;;printable;;auto' --case '(a|b)*c:x;;nft;;printable;;auto' \
     --case '[aie]:;;nft;;printable;;auto' --case '(a|b)*c:x;;dft;;printable;;auto' 2>&1 | grep -v amdgpu.ids
done
} | tee -a gpurun_out/one_ab.log
