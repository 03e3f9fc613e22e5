import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import gen
from image_generation_amd.modules import Encoder
from oracle import nets
for n, B in [(64, 128), (64, 129), (64, 130), (64, 260)]:
    params = gen.make_params(n, "encoder", 11 + n)
    enc = Encoder(n); enc.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in params.items()}); enc = enc.cuda().train()
    p = {k: (torch.from_numpy(np.array(v)).requires_grad_(True) if np.array(v).dtype == np.float32 and "running" not in k else torch.from_numpy(np.array(v))) for k, v in params.items()}
    x = torch.from_numpy(gen.make_images(B, 5)); gl = torch.randn(B, n, generator=torch.Generator().manual_seed(1))
    want = nets.encoder_forward(p, x, training=True); (want * gl).sum().backward()
    got = enc(x.cuda()); (got * gl.cuda()).sum().backward()
    print(n, B, "logits", float((got.detach().cpu() - want.detach()).abs().max()))
    for name, prm in enc.named_parameters():
        w = p[name].grad; g = prm.grad.cpu()
        print("   ", name, "relerr %.2e" % float((g - w).abs().max() / (w.abs().max() + 1e-30)), "scale %.2e" % float(w.abs().max()))
