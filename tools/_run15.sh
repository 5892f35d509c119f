timeout 900 python -m pytest tests/test_gpu_f16x3_range.py -q -m gpu -k "split_role" 2>&1 | grep -E "^E|assert|lean" | head -20
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "c5_fullsize" -s 2>&1 | grep -E "^C5|^E  |swapped" | head -20
