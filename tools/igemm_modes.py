"""The implicit-GEMM kernel alone in its three operand modes (float32 / f32x3 / bf16 inputs, dvg_set_conv_precision) on
c3-sized layer shapes: `python tools/igemm_modes.py` on an MI355X.  Times one launch (HIP events over 20 repeats), prices
it in 9-tap float32 FLOPs and checks the result against a float64 convolution of the same operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_generation_amd import _lib, dev


def run(M, Cin, Cout, L, x, w, reps=20):
    wp = torch.empty(9 * Cin * Cout * 2, device="cuda")
    out = dev.conv_igemm(x, w, 2, M, Cin, Cout, L, 9, 0, 0, wp=wp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dev.conv_igemm(x, w, 2, M, Cin, Cout, L, 9, 0, 0, wp=wp, repack=False)
    e1.record(); torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / reps * 1e3


for arg in [a for a in sys.argv[1:] if "=" in a]:
    name, value = arg.split("=")
    _lib.set_option(name, int(value))
    sys.argv.remove(arg)
shapes = [(128, 128, 2, 32768), (128, 128, 3, 8192), (64, 128, 3, 32768), (128, 64, 3, 32768), (64, 32, 4, 32768), (512, 128, 1, 32768)]
if len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
for Cin, Cout, L, imgs in shapes:
    M = imgs << (2 * L)
    if M * max(Cin, Cout) * 4 >= 2**32: continue
    torch.manual_seed(0)
    x = torch.randn(M, Cin, device="cuda"); w = torch.randn((Cin, Cout, 3, 3), device="cuda") / (3 * Cin**0.5)
    fl = 2.0 * M * Cin * Cout * 9
    # float64 truth on a slice of images (ConvTranspose2d fwd, mode 2)
    ni = 8
    xi = dev.morton_to_nchw(x[: ni << (2 * L)], ni, Cin, 1 << L).double()
    want = torch.nn.functional.conv_transpose2d(xi, w.double(), padding=1)
    line = f"Cin={Cin:3d} Cout={Cout:3d} L={L} M={M:8d} "
    for mode in ("f32", "f32x3", "bf16"):
        _lib.set_conv_precision(mode)
        o, t = run(M, Cin, Cout, L, x, w)
        got = dev.morton_to_nchw(o[: ni << (2 * L)], ni, Cout, 1 << L).double()
        err = float((got - want).abs().max() / want.abs().max())
        line += f" | {mode} {t:7.1f} us {fl/t/1e6:6.1f} TF/s err {err:.1e}"
    _lib.set_conv_precision("f32")
    print(line, flush=True)
