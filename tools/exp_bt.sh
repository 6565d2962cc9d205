#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/kbench.py --bytes 268435456 --steps 3 --case 'a(a|b|c|d|e|f|g|h){12}c:x;;nft;;printable;;backtrack' 2>&1 | grep "^pattern\|rror" | cut -c1-80,100-190
timeout 900 python -m pytest tests/test_backtrack.py -q -m gpu -x 2>&1 | tail -3
