timeout 900 python -m pytest tests/test_gpu_conv_variants.py -q -m gpu -k "pose7x7_192to128_c4" 2>&1 | tail -12
TERRAN_AMD_NO_ACT_SCALES=1 timeout 900 python -m pytest tests/test_gpu_conv_variants.py -q -m gpu -k "pose7x7_192to128_c4" 2>&1 | tail -6
