cd ${GRAFT_REPO_ROOT:-/root/repo}
for lb in 0 16384 32768; do
  if [ $lb = 0 ]; then env=""; else env="TRRE_LANE_BYTES=$lb"; fi
  echo "## dict lane_bytes=$lb"
  env $env python tools/kbench.py --dict 1000 --engine dft --bytes 8589934592 --steps 5 2>&1 | grep pattern
done
for lb in 0 16384; do
  if [ $lb = 0 ]; then env=""; else env="TRRE_LANE_BYTES=$lb"; fi
  echo "## guided / expand lane_bytes=$lb"
  env $env python tools/kbench.py --bytes 8589934592 --steps 5 --case '(a|b)*c:x;;nft;;printable;;auto' --case 'a:xyz;;dft;;printable;;auto' 2>&1 | grep pattern
done
