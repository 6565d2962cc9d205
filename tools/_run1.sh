#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for t in 0 256 1024; do
echo "== TRRE_SPLICE_THREADS=$t"
TRRE_SPLICE_THREADS=$t python tools/kbench.py --dict 1000 --engine dft --bytes 8589934592 --steps 4 2>&1 | grep "pattern" | cut -c1-30,100-220 | tail -1
done
