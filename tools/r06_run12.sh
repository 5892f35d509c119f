export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_variants.py -m gpu -x -q -k "rf_dwpw" 2>&1 | tail -12
