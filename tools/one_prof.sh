#!/bin/bash
# A/B knobs of the one-pass kernel, 8 GiB
for pr in 0 1; do
echo "== pairs $pr"
TRRE_ONE_PAIRS=$pr timeout 300 python tools/kbench.py --bytes 8589934592 --steps 10 --sum --case 'a:xyz;;dft;;printable;;auto' --case '(a|b)*c:x;;nft;;printable;;auto' --case ' +: ;;nft;;printable;;auto' 2>&1 | grep -E "pattern="
done
echo "== pairs 0, look 16"
TRRE_ONE_LOOK=16 timeout 300 python tools/kbench.py --bytes 8589934592 --steps 10 --sum --case 'a:xyz;;dft;;printable;;auto' 2>&1 | grep -E "pattern="
