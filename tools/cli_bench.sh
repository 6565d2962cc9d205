#!/bin/bash
# The command-line work-alike end to end on a file in /dev/shm (where the stages' time goes: TRRE_TRACE=1).
#   tools/cli_bench.sh [bytes]
n=${1:-2147483648}
cd ${GRAFT_REPO_ROOT:-/root/repo}
f=/dev/shm/trre_cli_bench.txt
python - <<PY
import sys
sys.path.insert(0, "tools")
import torch, corpora
d = corpora.printable_lines($n, corpora.SEED0 + 2, torch.device("cuda", 0))
d.cpu().numpy().tofile("$f")
PY
ls -la $f
for env in "" "TRRE_CLI_BLOCK=67108864"; do
  for rep in 1 2; do
    echo "## $env"; ( time env TRRE_TRACE=1 $env trre_amd/bin/trre_dft '[a:A-z:Z]' $f > /dev/null ) 2>&1 | grep -v "^$\|^user\|^sys"
  done
done
echo "--- pipe"
( time bash -c "cat $f | TRRE_TRACE=1 trre_amd/bin/trre_dft '[a:A-z:Z]' > /dev/null" ) 2>&1 | grep -v "^$\|^user\|^sys"
echo "--- reference read speed: cat to /dev/null, dd"
( time cat $f > /dev/null ) 2>&1 | grep real
rm -f $f
