"""Decoder forward + backward alone at c3's batch (no sampler, no MMD, nothing of the other network beside it): run under
rocprofv3 by tools/nets_alone.sh for per-kernel durations.  OPTS=side_stream=0 in the environment also takes the library's
own weight-gradient side stream away, so that every kernel is timed with the chip to itself."""
import sys, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen
from image_generation_amd.modules import Decoder
from image_generation_amd import _lib
for kv in os.environ.get('OPTS','').split(','):
    if kv: _lib.set_option(kv.split('=')[0], int(kv.split('=')[1]))
n, B, R = 512, 4096, 8
params = gen.make_params(n, "decoder", 11 + n)
dec = Decoder(n); dec.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in params.items()}); dec = dec.cuda().train()
spins = (torch.rand(B, R, n, device="cuda") < 0.5).float() * 2 - 1
spins.requires_grad_(True)
for it in range(8):
    for p in dec.parameters(): p.grad = None
    out = dec(spins)
    go = torch.randn_like(out)
    out.backward(go)
torch.cuda.synchronize()
print("done", out.shape)
