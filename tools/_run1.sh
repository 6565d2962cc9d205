python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dictionar" 2>&1 | tail -3
bash tools/prof_dict4.sh r04c 2>&1 | grep "k_fb\|GB/s" | cut -c1-160
