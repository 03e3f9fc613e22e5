#!/usr/bin/env python3
"""Times the decoder's two Upsample(x2) + 3x3 layers (128 -> 64 @4x4, 64 -> 32 @8x8; c3: 32768 images) in the F(4x4,3x3) form
with 25 of 36 positions (csrc/conv_wino4.hip, UM = 1) against the folded direct GEMM, alone on the chip.
    python tools/wino4_dec_bench.py [N]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import image_generation_amd  # noqa: E402,F401
from image_generation_amd import dev  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768


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


for name, Cin, Cout, side in (("dec 128->64 @4x4", 128, 64, 4), ("dec 64->32 @8x8", 64, 32, 8)):
    L, M = side.bit_length() - 1, N * side * side
    xs = torch.randn(M // 4, Cin, device="cuda")
    w = torch.randn(Cin, Cout, 3, 3, device="cuda") / 30
    t4 = min(timeit(lambda: dev.conv_wino4(xs, w, 2, M, Cin, Cout, L, um=1)) for _ in range(2))
    td = min(timeit(lambda: dev.conv_igemm(xs, w, 2, M, Cin, Cout, L, ups=1)) for _ in range(2))
    g4 = 2.0 * (M / 16) * 25 * Cin * Cout / 1e9
    print(f"{name}: F(4x4) behind the upsample {t4:7.1f} us ({g4 / t4 * 1e3 / 157.3:.2f} of the f32 peak on its 25 position GEMMs)   direct (upsample fused) {td:7.1f} us")
    gy = torch.randn(M, Cout, device="cuda")
    t4 = min(timeit(lambda: dev.conv_wino4(gy, w, 3, M, Cout, Cin, L, um=2)) for _ in range(2))
    td = min(timeit(lambda: dev.conv_igemm(gy, w, 3, M, Cout, Cin, L, poolsum=1)) for _ in range(2))
    print(f"{name}: its data gradient, F(4x4) {t4:7.1f} us ({g4 / t4 * 1e3 / 157.3:.2f})   direct (2x2 sum fused) {td:7.1f} us")
