#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -x -q -m gpu -k "stack_limit or divergence" 2>&1 | tail -3
