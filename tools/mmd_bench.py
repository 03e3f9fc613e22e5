#!/usr/bin/env python3
"""Times the MMD call (dvg_mmd_fwd_bwd) alone, per kernel, with the library's HIP-event profiler.

    python tools/mmd_bench.py [nx ny d] [--w128 0|1]

Default shape = c3's (32768, 256, 512).  Prints the per-kernel averages (prep / distance sum / pair kernel / final) and
the pair kernel's rate in bf16-equivalent TFLOP/s (int8 FLOPs x 0.5 + bf16 FLOPs: bench.py's pricing)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
import torch  # noqa: E402

import image_generation_amd  # noqa: E402,F401
from image_generation_amd import _lib, functional as F  # noqa: E402

if "--w128" in sys.argv:
    _lib.set_option("mmd_w128", int(sys.argv[sys.argv.index("--w128") + 1]))

nx, ny, d = (int(v) for v in args[:3]) if len(args) >= 3 else (32768, 256, 512)
g = torch.Generator().manual_seed(0)
x = ((torch.rand(nx, d, generator=g) < 0.4).float() * 2 - 1).cuda()
y = ((torch.rand(ny, d, generator=g) < 0.55).float() * 2 - 1).cuda()
L = _lib.lib()
names = [L.dvg_prof_kernel_name(i).decode() for i in range(L.dvg_prof_num_kernels())]
for _ in range(3):
    F.mmd_loss_and_grad(x, y)
torch.cuda.synchronize()
L.dvg_prof_reset()
L.dvg_prof_enable((1 << len(names)) - 1)
reps = 10
call = (lambda: F.mmd_loss(x, y)) if "--nograd" in sys.argv else (lambda: F.mmd_loss_and_grad(x, y))  # --nograd: the loss-only walk
for _ in range(reps):
    with torch.no_grad():
        call()
torch.cuda.synchronize()
L.dvg_prof_enable(0)
tot = 0.0
for i, nm in enumerate(names):
    ms, cnt, work = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    L.dvg_prof_query(i, ctypes.byref(ms), ctypes.byref(cnt))
    L.dvg_prof_query_work(i, ctypes.byref(work))
    if cnt.value:
        tot += ms.value / reps
        extra = f"  {work.value / (ms.value * 1e-3) / 1e12:8.1f} TFLOP/s-equivalent" if work.value and nm == "mmd_pm1" else ""
        print(f"{nm:16s} {ms.value / reps * 1e3:10.1f} us/call  ({cnt.value // reps} launches){extra}")
print(f"sum of kernels   {tot * 1e3:10.1f} us/call   shape ({nx}, {ny}, {d})  mmd_w128={_lib.get_option('mmd_w128')}")
