import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import gen
from image_generation_amd.modules import Encoder
n, B = 64, 128
params = gen.make_params(n, "encoder", 11 + n)
x = torch.from_numpy(gen.make_images(B, 5)).cuda(); gl = torch.randn(B, n, generator=torch.Generator().manual_seed(1)).cuda()
res = []
for rep in range(3):
    enc = Encoder(n); enc.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in params.items()}); enc = enc.cuda().train()
    got = enc(x); (got * gl).sum().backward(); torch.cuda.synchronize()
    res.append({k: p.grad.clone() for k, p in enc.named_parameters()})
for k in res[0]:
    d1 = float((res[0][k] - res[1][k]).abs().max()); d2 = float((res[0][k] - res[2][k]).abs().max())
    print(k, "run-to-run diff", d1, d2, "scale", float(res[0][k].abs().max()))
