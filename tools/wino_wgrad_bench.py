#!/usr/bin/env python3
"""The Winograd weight-gradient kernel (csrc/conv_wino_wgrad.hip) against the direct 3x3 weight-gradient kernels on the
encoder's three layers at c3's batch (and at B = 1024): time per call (slab reduce included), rate in direct-form FLOPs,
and the distance of both from a float64 weight gradient.    python tools/wino_wgrad_bench.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_generation_amd import _lib, dev

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, Cin, Cout, L in (("layer 1", 32, 64, 4), ("layer 2", 64, 128, 3), ("layer 3", 128, 512, 2)):
    M = B << (2 * L)
    torch.manual_seed(0)
    x = torch.randn(M, Cin, device="cuda"); dy = torch.randn(M, Cout, device="cuda")
    shape = (Cout, Cin, 3, 3)
    gd = dev.conv_wgrad(x, dy, 0, shape, M, Cin, Cout, L)
    gw = dev.conv_wino_wgrad(x, dy, 0, shape, M, Cin, Cout, L)
    # float64 reference on a slice of the batch would not be the same sum: use stock conv weight gradient in float64 on a
    # subset of output channels / all pixels via unfold-free formula: dW[co,ci,r,s] = sum_m dy[m,co] x[nbr(m,r,s),ci]
    side = 1 << L
    xn = dev.morton_to_nchw(x, B, Cin, side).double(); dyn = dev.morton_to_nchw(dy, B, Cout, side).double()
    ref = torch.nn.grad.conv2d_weight(xn[:, :8], (Cout, 8, 3, 3), dyn, padding=1) if False else None
    nb = min(B, 256)
    ref = torch.nn.grad.conv2d_weight(xn[:nb], shape, dyn[:nb], padding=1)
    gd_s = dev.conv_wgrad(dev.nchw_to_morton(xn[:nb].float()), dev.nchw_to_morton(dyn[:nb].float()), 0, shape, nb << (2 * L), Cin, Cout, L)
    gw_s = dev.conv_wino_wgrad(dev.nchw_to_morton(xn[:nb].float()), dev.nchw_to_morton(dyn[:nb].float()), 0, shape, nb << (2 * L), Cin, Cout, L)
    rel = lambda a: float((a.double() - ref).norm() / ref.norm())
    t_d = timeit(lambda: dev.conv_wgrad(x, dy, 0, shape, M, Cin, Cout, L))
    gf = 2.0 * M * Cin * Cout * 9 / 1e9
    ge = 2.0 * (M / 4) * 16 * Cin * Cout / 1e9  # executed GFLOP of the Winograd form
    print(f"{name} M={M:8d} {Cin:4d}->{Cout:4d}  direct (grid sized for 256 CUs) {t_d:7.1f} us ({gf / t_d * 1e3:6.1f} TFLOP/s direct-form)"
          f"   wino vs direct {float((gw - gd).norm() / gd.norm()):.1e}   vs float64 (B={nb}): wino {rel(gw_s):.1e} direct {rel(gd_s):.1e}")
    # the Winograd form at the CU budget the training step gives it (128: WINO_CUS_ENC_WGRAD of csrc/conv.h, beside the data gradient)
    # and on the whole chip; interleaved rounds in one process (minimum of 2); slab reduce included
    ts = {}
    for rnd in range(2):
        for cus in (128, 256):
            t = timeit(lambda: dev.conv_wino_wgrad(x, dy, 0, shape, M, Cin, Cout, L, cus=cus))
            ts[cus] = min(ts.get(cus, 1e30), t)
    for cus in (128, 256):
        print(f"        wino, grid sized for {cus:3d} CUs: {ts[cus]:7.1f} us ({gf / ts[cus] * 1e3:6.1f} TFLOP/s direct-form, {ge / ts[cus] * 1e3 / 157.3:.2f} of the f32 peak executed)")
