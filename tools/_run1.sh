python tools/nul_bench.py 2>&1 | tail -3
python tools/nul_bench.py --bytes $((8<<30)) --steps 3 2>&1 | tail -3
TRRE_NO_NUL_REPAIR=1 python tools/nul_bench.py --bytes $((8<<30)) --steps 3 2>&1 | tail -3
python tools/kbench.py --case "(cat:dog|dog:cat);;nft;;catdog;;auto" --steps 5 2>&1 | tail -1
python tools/kbench.py --case "(cat:dog|dog:cat);;nft;;catdog;;auto" --steps 5 --out-mis 5 2>&1 | tail -1
TRRE_LPW_ALIGNED_ONLY=1 python tools/kbench.py --case "(cat:dog|dog:cat);;nft;;catdog;;auto" --steps 5 --out-mis 5 2>&1 | tail -1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nul or relaunch or unaligned or golden or every_kernel" 2>&1 | tail -3
