run() { python tools/kbench.py --dict 1000 --engine dft --steps 10 2>&1 | tail -1 | cut -c100-; }
echo "default lib (win 2304, literals from memory, 512 thr)"; run
echo "default lib, literals in LDS"; TRRE_SPLICE_LIT_LDS=1 run
echo "win 2176 grow 384, literals from memory (3 WGs/CU)"; TRRE_LIB_PATH=$PWD/trre_amd/lib8/libtrre_mi355x.so run
echo "win 2176, literals in LDS"; TRRE_SPLICE_LIT_LDS=1 TRRE_LIB_PATH=$PWD/trre_amd/lib8/libtrre_mi355x.so run
TRRE_LIB_PATH=$PWD/trre_amd/lib8/libtrre_mi355x.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dictionar" 2>&1 | tail -2
