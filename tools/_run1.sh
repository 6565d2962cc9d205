#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stack_limit or match_mode" 2>&1 | tail -30
