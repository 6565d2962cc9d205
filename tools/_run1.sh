#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lb in 0 8192 16384 32768; do
echo "== TRRE_LANE_BYTES=$lb"
TRRE_LANE_BYTES=$lb python tools/kbench.py --case 'a:xyz;;dft;;printable;;auto' --case ' +: ;;nft;;printable;;auto' --case '(cat:dog|dog:cat);;nft;;catdog;;stream_gen' --bytes 8589934592 --steps 3 2>&1 | grep pattern | cut -c1-60,100-200
done
