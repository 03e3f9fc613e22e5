#!/usr/bin/env python3
"""The Winograd weight-gradient kernel on the decoder's Upsample(x2) + 3x3 layers (9 of 16 transform positions) against the
folded direct kernel and float64:   python tools/wino_wgrad_ups_check.py [N]"""
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
MODE_CONVT_FWD = 2
for name, Cin, Cout, L in (("decoder layer 1", 128, 64, 2), ("decoder layer 2", 64, 32, 3)):
    M = N << (2 * L)
    torch.manual_seed(0)
    xs = torch.randn(M // 4, Cin, device="cuda"); dy = torch.randn(M, Cout, device="cuda")
    shape = (Cin, Cout, 3, 3)  # ConvTranspose2d layout
    gd = dev.conv_wgrad(xs, dy, MODE_CONVT_FWD, shape, M, Cin, Cout, L, ups=1)
    gw = dev.conv_wino_wgrad(xs, dy, MODE_CONVT_FWD, shape, M, Cin, Cout, L, ups=1)
    nb = min(N, 512)
    side = 1 << L
    xn = dev.morton_to_nchw(xs[: nb * side * side // 4], nb, Cin, side // 2).double()
    dyn = dev.morton_to_nchw(dy[: nb * side * side], nb, Cout, side).double()
    up = torch.nn.functional.interpolate(xn, scale_factor=2, mode="nearest")
    w = torch.zeros(shape, dtype=torch.float64, device="cuda", requires_grad=True)
    out = torch.nn.functional.conv_transpose2d(up, w, padding=1)
    (out * dyn).sum().backward()
    ref = w.grad
    gd_s = dev.conv_wgrad(xs[: nb * side * side // 4].contiguous(), dy[: nb * side * side].contiguous(), MODE_CONVT_FWD, shape, nb << (2 * L), Cin, Cout, L, ups=1)
    gw_s = dev.conv_wino_wgrad(xs[: nb * side * side // 4].contiguous(), dy[: nb * side * side].contiguous(), MODE_CONVT_FWD, shape, nb << (2 * L), Cin, Cout, L, ups=1)
    rel = lambda a: float((a.double() - ref).norm() / ref.norm())
    t_d = timeit(lambda: dev.conv_wgrad(xs, dy, MODE_CONVT_FWD, shape, M, Cin, Cout, L, ups=1))
    t_w = timeit(lambda: dev.conv_wino_wgrad(xs, dy, MODE_CONVT_FWD, shape, M, Cin, Cout, L, ups=1))
    print(f"{name} M={M:8d} {Cin:4d}->{Cout:4d}  wino-9 {t_w:8.1f} us   direct (folded) {t_d:8.1f} us   wino vs direct {float((gw - gd).norm() / gd.norm()):.1e}"
          f"   vs float64 (N={nb}): wino {rel(gw_s):.1e} direct {rel(gd_s):.1e}")
