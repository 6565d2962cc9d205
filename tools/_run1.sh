#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python tools/_run2.py 2>&1 | tail -12
echo "== guard off"
TRRE_NO_STACK_GUARD=1 timeout 600 python tools/_run2.py 2>&1 | tail -8
