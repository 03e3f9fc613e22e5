#!/usr/bin/env python3
"""Times the Winograd F(4x4,3x3) weight-gradient kernel (csrc/conv_wino4_wgrad.hip) against the F(2x2,3x3) one
(csrc/conv_wino_wgrad.hip) and the direct kernel on the encoder's three 3x3 layers at batch B (default 4096: c3), on the 128
CUs the training step gives the launch and on the whole chip.

    python tools/wino4_wgrad_bench.py [B]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import image_generation_amd  # noqa: E402,F401
from image_generation_amd import dev  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
shapes = [("L1", 32, 64, 16), ("L2", 64, 128, 8), ("L3", 128, 512, 4)]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"B = {B}; us per weight gradient (kernel + slab reduction), minimum of two interleaved rounds of 10")
for name, Cin, Cout, side in shapes:
    if ONLY and name != ONLY:
        continue
    L, M = side.bit_length() - 1, B * side * side
    x = torch.randn(M, Cin, device="cuda")
    dy = torch.randn(M, Cout, device="cuda")
    shape = (Cout, Cin, 3, 3)
    ts = {}
    for rnd in range(2):
        for key, fn in (("4/128", lambda: dev.conv_wino4_wgrad(x, dy, 0, shape, M, Cin, Cout, L, cus=128)),
                        ("4/256", lambda: dev.conv_wino4_wgrad(x, dy, 0, shape, M, Cin, Cout, L, cus=256)),
                        ("2/128", lambda: dev.conv_wino_wgrad(x, dy, 0, shape, M, Cin, Cout, L, cus=128)),
                        ("2/256", lambda: dev.conv_wino_wgrad(x, dy, 0, shape, M, Cin, Cout, L, cus=256)),
                        ("d", lambda: dev.conv_wgrad(x, dy, 0, shape, M, Cin, Cout, L))):
            ts[key] = min(ts.get(key, 1e30), timeit(fn))
    g4 = 2.0 * (M / 16) * 36 * Cin * Cout / 1e9
    g2 = 2.0 * (M / 4) * 16 * Cin * Cout / 1e9
    print(f"{name} M={M:8d} {Cin:4d}->{Cout:4d}  F(4x4) {ts['4/128']:7.1f} us on 128 CUs, {ts['4/256']:7.1f} on 256 ({g4 / ts['4/256'] * 1e3 / 157.3:.2f} of the f32 peak executed)   "
          f"F(2x2) {ts['2/128']:7.1f} / {ts['2/256']:7.1f} ({g2 / ts['2/256'] * 1e3 / 157.3:.2f})   direct {ts['d']:7.1f}   F(4x4)/F(2x2) {ts['4/128'] / ts['2/128']:.2f} / {ts['4/256'] / ts['2/256']:.2f}")
