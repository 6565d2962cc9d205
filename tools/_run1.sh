#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k random_patterns --durations=2 2>&1 | tail -5
