import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import gen
from image_generation_amd.modules import Encoder
from oracle import nets
def mk(params, dt):
    out = {}
    for k, v in params.items():
        t = torch.from_numpy(np.array(v))
        if t.dtype == torch.float32:
            t = t.to(dt)
            if "running" not in k: t.requires_grad_(True)
        out[k] = t
    return out
for n, B in [(64, 128), (64, 130)]:
    params = gen.make_params(n, "encoder", 11 + n)
    enc = Encoder(n); enc.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in params.items()}); enc = enc.cuda().train()
    x = torch.from_numpy(gen.make_images(B, 5)); gl = torch.randn(B, n, generator=torch.Generator().manual_seed(1))
    p32 = mk(params, torch.float32); p64 = mk(params, torch.float64)
    w32 = nets.encoder_forward(p32, x, training=True); (w32 * gl).sum().backward()
    w64 = nets.encoder_forward(p64, x.double(), training=True); (w64 * gl.double()).sum().backward()
    got = enc(x.cuda()); (got * gl.cuda()).sum().backward()
    for name, prm in enc.named_parameters():
        t = p64[name].grad; s = float(t.abs().max()) + 1e-30
        print(n, B, name, "gpu-vs-f64 %.2e" % (float((prm.grad.cpu().double() - t).abs().max()) / s), "cpu32-vs-f64 %.2e" % (float((p32[name].grad.double() - t).abs().max()) / s))
