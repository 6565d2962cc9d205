#!/bin/bash
# the memoryless one-pass kernel (map_block.hpp, TRRE_MAPGEN=1) against the count / emit pair: rates and output checksums, 1 GiB and 8 GiB
for B in 1073741824 8589934592; do
for env in "TRRE_MAPGEN=0" "TRRE_MAPGEN=1" "TRRE_MAPGEN=1 TRRE_MAPGEN_WINDOW=32768"; do
  echo "== $env  bytes $B"
  env $env TRRE_TRACE=1 timeout 300 python tools/kbench.py --bytes $B --steps 10 --sum --case 'a:xyz;;dft;;printable;;auto' --case '[aie]:;;nft;;printable;;auto' --case '(a:xyz|e:)|.:uv;;dft;;printable;;auto' 2>&1 | grep -E "pattern=|void|rror"
done
done
TRRE_MAPGEN=1 TRRE_MAPGEN_PROF=1 timeout 300 python tools/kbench.py --bytes 8589934592 --steps 3 --case "a:xyz;;dft;;printable;;auto" 2>&1 | grep -E "memoryless" | tail -1
