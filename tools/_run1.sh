python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dictionar" 2>&1 | tail -3
for t in 1024 512 256; do TRRE_SPLICE_THREADS=$t python tools/kbench.py --dict 1000 --engine dft --steps 10 2>&1 | tail -1; done
TRRE_EMIT_DBG=1 python tools/kbench.py --dict 1000 --engine dft --steps 10 2>&1 | tail -1
