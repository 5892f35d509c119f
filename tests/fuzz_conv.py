"""Randomised conv cases through the checker of tests/test_gpu_conv.py (run on the GPU box).

    python tests/fuzz_conv.py [n_cases] [seed]
Shapes the networks never use are drawn on purpose: 1x1 .. 33x47 maps, batch 1..5, channel counts around the
32 / 64 / 128 tile edges, channel slices, strides, all epilogue variants.  Prints the failing case dicts."""
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib                      # noqa: E402
from tests.test_gpu_conv import test_conv      # noqa: E402


def draw(rng):
    k = int(rng.choice([1, 3, 3, 7]))
    c1 = int(rng.choice([4, 8, 16, 24, 32, 64, 96, 128, 160, 192, 256]))
    cout = int(rng.choice([4, 8, 19, 20, 32, 38, 60, 64, 100, 128, 132, 192, 256, 320]))
    case = dict(c1=c1, cout=cout, k=k, n=int(rng.integers(1, 6)), h=int(rng.integers(1, 34)), w=int(rng.integers(1, 48)))
    if rng.random() < 0.12:                      # grouped conv: 128-channel output tiles inside one group
        case.update(c1=int(rng.choice([64, 256, 512])), cout=int(rng.choice([256, 512])), groups=2, act=int(rng.choice([0, 1, 2])))
        if k == 7:
            case['halo'] = 3
        if k == 1 and case['c1'] == 64:          # one K slab per group: rejected by the planner (needs >= 2)
            case['c1'] = 256
        return case
    if k > 1 and rng.random() < 0.3:
        case['stride'] = 2
    elif k == 1 and rng.random() < 0.2:
        case['stride'] = 2
        case['pad'] = 0
    if k == 3 and rng.random() < 0.3:
        case['halo'] = 3
    if c1 >= 64 and rng.random() < 0.3:
        used = int(rng.choice([16, 32, c1 // 2]))
        case['cin_used'] = used
        case['in_off'] = int(rng.choice([0, 32, c1 - used])) // 4 * 4
        if case['in_off'] + used > c1:
            case['in_off'] = 0
    if cout % 4:
        case['cout_p'] = (cout + 3) // 4 * 4
    if rng.random() < 0.25:
        tot = ((cout + 3) // 4 * 4) + int(rng.choice([32, 64, 128]))
        case['out_total'] = tot
        case['out_off'] = int(rng.choice([0, 32, tot - (cout + 3) // 4 * 4])) // 4 * 4
    case['act'] = int(rng.choice([0, 1, 2]))
    if k == 3 and case.get('stride', 1) == 1 and 'cin_used' not in case and rng.random() < 0.25:
        case['in_affine'] = True                 # folded input affine + border-class biases (no second output with it)
        return case
    if rng.random() < 0.3 and 'out_total' not in case:
        case['res'] = True
    if rng.random() < 0.2 and 'out_total' not in case:
        case['out2'] = True
    return case


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ctx = lib.Context(0)
    bad = 0
    for i in range(n):
        case = draw(rng)
        for prec in ('f32', 'f16x3', 'bf16x3', 'f16', 'f16x2'):
            if prec == 'f16' and case.get('groups', 1) > 1:      # the single-half mode has no grouped convs
                continue
            try:
                test_conv(ctx, case, prec)
            except Exception as e:          # noqa: BLE001
                bad += 1
                print('FAIL', prec, case)
                print('   ', ''.join(traceback.format_exception_only(type(e), e)).strip()[:600])
    print('%d cases x 5 precisions, %d failures' % (n, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
