"""A/B of the 3x3 weight-gradient kernel's two staging forms (register-staged vs LDS-DMA, option wgrad_dma) on c3- and
c2-sized layer shapes: `PYTHONPATH=. python tools/wgrad_ab.py` on an MI355X (times include the slab reduce pass)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_generation_amd import _lib, dev

def run(x, dy, M, Cin, Cout, L, ups, dma, reps=10):
    _lib.set_option("wgrad_dma", int(dma))
    shape = (Cin, Cout, 3, 3)
    g = dev.conv_wgrad(x, dy, 2, shape, M, Cin, Cout, L, ups=ups)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dev.conv_wgrad(x, dy, 2, shape, M, Cin, Cout, L, ups=ups)
    e1.record(); torch.cuda.synchronize()
    return g, e0.elapsed_time(e1) / reps * 1e3

shapes = [(128, 128, 2, 32768, 0), (128, 128, 3, 8192, 0), (64, 128, 3, 32768, 0), (128, 64, 3, 32768, 1), (64, 64, 4, 8192, 0),
          (32, 64, 4, 8192, 0), (64, 32, 4, 8192, 1), (128, 128, 1, 32768, 0),
          (128, 128, 2, 2048, 0), (64, 128, 3, 2048, 0), (128, 64, 3, 2048, 1), (128, 128, 1, 2048, 0)]
for Cin, Cout, L, imgs, ups in shapes:
    M = imgs << (2 * L)
    if M * max(Cin, Cout) * 4 >= 2**32: continue
    torch.manual_seed(0)
    x = torch.randn(M // 4 if ups else M, Cin, device="cuda"); dy = torch.randn(M, Cout, device="cuda")
    g0, t0 = run(x, dy, M, Cin, Cout, L, ups, "0")
    g1, t1 = run(x, dy, M, Cin, Cout, L, ups, "1")
    fl = 2.0 * M * Cin * Cout * 9
    err = float((g0 - g1).abs().max() / g0.abs().max())
    print(f"Cin={Cin:3d} Cout={Cout:3d} L={L} ups={ups} M={M:8d}  reg {t0:8.1f} us {fl/t0/1e6:6.1f} TF/s   dma {t1:8.1f} us {fl/t1/1e6:6.1f} TF/s   x{t0/t1:5.2f}  maxdiff {err:.2e}")
