#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/kbench.py --case 'a(a|b|c|d|e|f|g|h){12}c:x;;nft;;printable;;auto' --case '(cat:dog|dog:cat);;nft;;catdog;;backtrack' --case '(a|b)*c:x;;nft;;printable;;backtrack' --bytes 268435456 --steps 3 2>&1 | grep pattern | cut -c1-40,100-200
