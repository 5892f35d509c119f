for g in 22 16 12 8 4 0; do python tools/amax_debug.py arcface f16x3 --stem-gain $g 2>&1 | grep -E "ops that|range check" ; done
