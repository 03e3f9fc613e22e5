#!/usr/bin/env python3
"""Where the F(4x4,3x3) kernel's time goes: launch time against the number of 4-channel chunks per tile block (Cin / 4) at
a fixed number of tile blocks per workgroup -- slope = time per chunk, intercept = time per tile block beside its chunks.

    python tools/wino4_sweep.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import image_generation_amd  # noqa: E402,F401
from image_generation_amd import dev  # noqa: E402


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


for side, B, Cout in ((16, 4096, 64), (8, 16384, 64), (4, 65536, 64), (16, 4096, 32)):
    L, M = side.bit_length() - 1, B * side * side
    nblk, ny = M // 1024, Cout // 32
    per_wg = nblk * ny / 256
    rows = []
    for Cin in (8, 32, 128, 256):
        x = torch.randn(M, Cin, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") / 30
        t = min(timeit(lambda: dev.conv_wino4(x, w, 0, M, Cin, Cout, L)) for _ in range(2))
        rows.append((Cin // 4, t))
        del x
    (c0, t0), (c1, t1) = rows[-3], rows[-1]
    slope = (t1 - t0) / (c1 - c0) / per_wg
    print(f"side {side:2d} Cout {Cout:3d} tile blocks per workgroup {per_wg:.0f}: " + "  ".join(f"{c} chunks {t:7.1f} us" for c, t in rows) +
          f"   -> {slope:.2f} us per chunk (ideal 0.98 at 2.4 GHz), {(t0 / per_wg - c0 * slope):.1f} us per tile block beside its chunks")
