#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python tools/kbench.py --case 'a:xyz;;dft;;printable;;auto' --case '(a|b)*c:x;;nft;;printable;;auto' --case '(.:x)*.*;;nft;;printable;;auto' --case ' +: ;;nft;;printable;;auto' --bytes 8589934592 --steps 5 2>&1 | grep pattern | cut -c1-40,100-200; }
run; run
