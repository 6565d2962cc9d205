#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python tools/kbench.py --case '(a|b)*c:x;;nft;;printable;;auto' --case '(.:x)*.*;;nft;;printable;;auto' --case '(cat:dog|dog:cat);;nft;;catdog;;guided_lp' --case '[0-9]+:N;;nft;;printable;;auto' --bytes 8589934592 --steps 5 2>&1 | grep pattern | cut -c1-40,100-200; }
echo "== new rule"; run
echo "== 2048"; TRRE_LANE_BYTES=2048 run
