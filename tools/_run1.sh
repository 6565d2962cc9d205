python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dictionar" 2>&1 | tail -2
bash tools/prof_dict4.sh r04e 2>&1 | grep "k_fb\|GB/s" | cut -c1-160 | head -3
