"""A/B of the float32 implicit-GEMM kernel's two staging forms (register-staged vs LDS-DMA, option igemm_dma) on c3- and
c2-sized layer shapes: `PYTHONPATH=. python tools/igemm_ab.py` on an MI355X.  Also checks that both give the same answer."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_generation_amd import _lib, dev

def run(M, Cin, Cout, L, dma, x, w, reps=20):
    _lib.set_option("igemm_dma", int(dma))
    wp = torch.empty(9 * Cin * Cout * 2, device="cuda")
    out = dev.conv_igemm(x, w, 2, M, Cin, Cout, L, 9, 0, 0, wp=wp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dev.conv_igemm(x, w, 2, M, Cin, Cout, L, 9, 0, 0, wp=wp, repack=False)
    e1.record(); torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / reps * 1e3

shapes = [(128, 128, 2, 32768), (128, 128, 3, 8192), (64, 128, 3, 32768), (128, 64, 3, 32768), (64, 64, 4, 8192), (64, 32, 4, 32768),
          (128, 128, 2, 2048), (64, 128, 3, 2048), (128, 64, 3, 2048), (128, 128, 1, 2048), (64, 32, 4, 2048)]
for Cin, Cout, L, imgs in shapes:
    M = imgs << (2 * L)
    if M * max(Cin, Cout) * 4 >= 2**32: continue
    torch.manual_seed(0)
    x = torch.randn(M, Cin, device="cuda"); w = torch.randn((Cin, Cout, 3, 3), device="cuda") / (3 * Cin**0.5)
    o0, t0 = run(M, Cin, Cout, L, "0", x, w)
    o1, t1 = run(M, Cin, Cout, L, "1", x, w)
    fl = 2.0 * M * Cin * Cout * 9
    err = float((o0 - o1).abs().max() / o0.abs().max())
    print(f"Cin={Cin:3d} Cout={Cout:3d} L={L} M={M:8d}  reg {t0:8.1f} us {fl/t0/1e6:6.1f} TF/s   dma {t1:8.1f} us {fl/t1/1e6:6.1f} TF/s (9-tap FLOPs: position-major tiles execute fewer)   x{t0/t1:5.2f}  maxdiff {err:.2e}")
