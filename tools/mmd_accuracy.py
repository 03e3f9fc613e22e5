#!/usr/bin/env python3
"""How accurate is the MMD gradient at c3's size (32768 x 256 rows of 512 spins), by implementation?

For 96 sampled rows of x the gradient is evaluated in float64 on the device (the estimator's terms that involve a
sampled row, differentiated by autograd with the oracle's distance / kernel-factor functions: the same check as
tests/test_gpu_fullsize.py::test_mmd_c3_size_against_float64_on_sampled_rows) and compared with
  (a) the 128-row-block spin kernel (int8 Gram, 2 bf16 terms per weight, bf16 MFMA, float32 accumulate),
  (b) the 32-row-block spin kernel (3 bf16 terms),
  (c) the general float32 kernel (f32 MFMA = exact fmaf chain), reached by nudging one entry of y by one ulp.
Printed: max |error| relative to the largest |gradient| and relative to the largest |x-x part| + |x-y part|."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import image_generation_amd  # noqa: E402,F401
from image_generation_amd import _lib, functional as F  # noqa: E402
from oracle import plugin  # noqa: E402

nx, ny, d = 32768, 256, 512
g = torch.Generator().manual_seed(3)
x = ((torch.rand(nx, d, generator=g) < 0.4).float() * 2 - 1).cuda()
y = ((torch.rand(ny, d, generator=g) < 0.55).float() * 2 - 1).cuda()
z = torch.cat([x, y]).double()
N = nx + ny
dsum = torch.zeros((), dtype=torch.float64, device="cuda")
for r0 in range(0, N, 4096):
    dsum += plugin.pairwise_distance(z[r0:r0 + 4096], z, False).sum()
bws = (dsum / (N * N - N)) * plugin.kernel_factors(7, 2.0).to("cuda", torch.float64)
kern = lambda a_, b_: torch.exp(-plugin.pairwise_distance(a_, b_, False).unsqueeze(0) / bws.reshape(-1, 1, 1)).sum(0)  # noqa: E731
idx = torch.cat([torch.tensor([0, 31, 32, 127, 128, nx - 129, nx - 128, nx - 1]), torch.randint(0, nx, (88,), generator=g)]).unique().cuda()
xs = z[idx].clone().requires_grad_(True)
sel = torch.zeros(nx, dtype=torch.bool, device="cuda")
sel[idx] = True
kss = kern(xs, xs)
p_xx = (2.0 * kern(xs, z[:nx])[:, ~sel].sum() + kss.sum() - kss.trace()) / (nx * (nx - 1))
p_xy = -2.0 * kern(xs, z[nx:]).sum() / (nx * ny)
g_xx, = torch.autograd.grad(p_xx, xs, retain_graph=True)
g_xy, = torch.autograd.grad(p_xy, xs)
want = g_xx + g_xy
parts = float((g_xx.abs() + g_xy.abs()).max())
print(f"max |grad| {float(want.abs().max()):.3e}   max (|xx part| + |xy part|) {parts:.3e}")


def run(label, yy):
    _, grad = F.mmd_loss_and_grad(x, yy)
    err = (grad[idx].double() - want)
    print(f"{label:44s} max err {float(err.abs().max()):.3e} = {float(err.abs().max()) / float(want.abs().max()):.2e} of max|grad|,"
          f" {float(err.abs().max()) / parts:.2e} of the parts; mean signed err / parts {float(err.mean()) / parts:+.2e}")


_lib.set_option("mmd_w128", 1)
run("(a) 128-row-block spin kernel, 2 bf16 terms", y)
_lib.set_option("mmd_w128", 0)
run("(b) 32-row-block spin kernel, 3 bf16 terms", y)
y2 = y.clone()
y2[-1, -1] = torch.nextafter(y2[-1, -1], torch.tensor(0.0, device="cuda"))
run("(c) general float32 kernel (f32 MFMA)", y2)
