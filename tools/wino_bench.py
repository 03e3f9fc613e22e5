#!/usr/bin/env python3
"""Times the Winograd F(2x2,3x3) form (csrc/conv_wino.hip) against the direct implicit GEMM on the encoder's six 3x3 launches
(forward and data gradient of layers 1-3) at batch B (default 4096: c3).

    python tools/wino_bench.py [B]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import image_generation_amd  # noqa: E402,F401
from image_generation_amd import _lib, dev  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
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
for name, Cin, Cout, side, mode in shapes:
    L, M = side.bit_length() - 1, B * side * side
    x = torch.randn(M, Cin, device="cuda")
    # (weights in the checkpoint layout of the FORWARD layer: (Cout_fwd, Cin_fwd, 3, 3))
    w = torch.randn((Cout, Cin, 3, 3) if mode == 0 else (Cin, Cout, 3, 3), device="cuda") / 30
    ts = {}
    for rnd in range(2):  # interleaved rounds of the two forms in one process; the minimum of each is printed
        ts["w"] = min(ts.get("w", 1e30), timeit(lambda: dev.conv_wino(x, w, mode, M, Cin, Cout, L)))
        ts["d"] = min(ts.get("d", 1e30), timeit(lambda: dev.conv_igemm(x, w, mode, M, Cin, Cout, L)))
    gf = 2.0 * M * Cin * Cout * 9 / 1e9  # direct-form GFLOP of the launch
    ge = 2.0 * (M / 4) * 16 * Cin * Cout / 1e9  # executed GFLOP (16 position GEMMs per quad)
    print(f"{name:9s} M={M:8d} {Cin:4d}->{Cout:4d}  wino {ts['w']:7.1f} us ({ge / ts['w'] * 1e3:5.1f} TFLOP/s executed = {ge / ts['w'] * 1e3 / 157.3:.2f} of the f32 peak; "
          f"{gf / ts['w'] * 1e3:5.1f} TFLOP/s of direct-form FLOPs)   direct {ts['d']:7.1f} us ({gf / ts['d'] * 1e3:5.1f} TFLOP/s = {gf / ts['d'] * 1e3 / 157.3:.2f})")
