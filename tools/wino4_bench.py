#!/usr/bin/env python3
"""Times the Winograd F(4x4,3x3) form (csrc/conv_wino4.hip) against the F(2x2,3x3) form (csrc/conv_wino.hip) and the direct
implicit GEMM on the encoder's six 3x3 launches (forward and data gradient of layers 1-3) at batch B (default 4096: c3),
alone on the chip, and on the 128 CUs a training step gives the data gradients.

    python tools/wino4_bench.py [B]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import image_generation_amd  # noqa: E402,F401
from image_generation_amd import _lib, dev  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""   # e.g. "L1 fwd": that launch alone (PMC passes)
n = 512
shapes = [("L1 fwd", 32, 64, 16, 0), ("L2 fwd", 64, 128, 8, 0), ("L3 fwd", 128, n, 4, 0),
          ("L3 dgrad", n, 128, 4, 1), ("L2 dgrad", 128, 64, 8, 1), ("L1 dgrad", 64, 32, 16, 1)]


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


_lib.set_option("enc_wino", 1)
print(f"B = {B}; us per launch alone on the chip (minimum of two interleaved rounds of 10), pack launch included in every form")
for name, Cin, Cout, side, mode in shapes:
    if ONLY and name != ONLY:
        continue
    L, M = side.bit_length() - 1, B * side * side
    x = torch.randn(M, Cin, device="cuda")
    w = torch.randn((Cout, Cin, 3, 3) if mode == 0 else (Cin, Cout, 3, 3), device="cuda") / 30
    ts = {}
    for rnd in range(2):
        ts["4"] = min(ts.get("4", 1e30), timeit(lambda: dev.conv_wino4(x, w, mode, M, Cin, Cout, L)))
        ts["4h"] = min(ts.get("4h", 1e30), timeit(lambda: dev.conv_wino4(x, w, mode, M, Cin, Cout, L, cus=128)))
        ts["2"] = min(ts.get("2", 1e30), timeit(lambda: dev.conv_wino(x, w, mode, M, Cin, Cout, L)))
        ts["d"] = min(ts.get("d", 1e30), timeit(lambda: dev.conv_igemm(x, w, mode, M, Cin, Cout, L)))
    g4 = 2.0 * (M / 16) * 36 * Cin * Cout / 1e9  # executed GFLOP, F(4x4): 36 position GEMMs per tile
    g2 = 2.0 * (M / 4) * 16 * Cin * Cout / 1e9
    gd = 2.0 * M * Cin * Cout * 9 / 1e9
    print(f"{name:9s} M={M:8d} {Cin:4d}->{Cout:4d}  F(4x4) {ts['4']:7.1f} us ({g4 / ts['4'] * 1e3 / 157.3:.2f} of the f32 peak executed; on 128 CUs {ts['4h']:7.1f})   "
          f"F(2x2) {ts['2']:7.1f} us ({g2 / ts['2'] * 1e3 / 157.3:.2f})   direct {ts['d']:7.1f} us ({gd / ts['d'] * 1e3 / 157.3:.2f})   F(4x4)/F(2x2) {ts['4'] / ts['2']:.2f}")
