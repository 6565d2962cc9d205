#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/kbench.py --steps 5 --case '(a|b)*a(a|b){18}:x;;dft;;printable;;auto' --case '((a:x)*b)|((a:y)*c);;dft;;printable;;auto' 2>&1 | grep "^pattern" | cut -c1-70,100-190
timeout 600 python -m pytest tests/test_lazy.py tests/test_gpu_round5.py -q -m gpu -x 2>&1 | tail -3
