"""Throughput of one implicit-GEMM layer shape as the launch grows (blocks per CU): run on an MI355X with
`PYTHONPATH=. python tools/igemm_size_scaling.py`.  Source of the 70 / 83 / 94 / 98 / 101 TFLOP/s figures in DESIGN.md §8."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from image_generation_amd import dev
# scaling of one layer shape with the number of images: how does time grow with blocks per CU?
for Cin, Cout, L in [(128, 128, 1), (64, 128, 3)]:
    for imgs in [512, 1024, 2048, 4096, 8192, 16384]:
        M = imgs * (1 << (2 * L))
        x = torch.randn(M, Cin, device="cuda")
        w = torch.randn((Cin, Cout, 3, 3), device="cuda")
        wp = torch.empty(9 * Cin * Cout, device="cuda")
        dev.conv_igemm(x, w, 2, M, Cin, Cout, L, 9, 0, 0, wp=wp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dev.conv_igemm(x, w, 2, M, Cin, Cout, L, 9, 0, 0, wp=wp, repack=False)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * M * Cin * Cout * 9
        print(f"Cin={Cin} Cout={Cout} L={L} M={M:8d} blocks64={(M//64)*(Cout//64):6d} {us:8.1f} us {fl/us/1e6:6.1f} TF/s")
