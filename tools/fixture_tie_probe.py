"""Encoder forward of step 0 of the 12-step fixture (tests/golden/step_n64.npz: seed-909 images, the wrapper's seeded
initialisation) under one operand mode; writes the BatchNorm outputs z of layers 1-3 (the max-pool inputs) to
/tmp/tie_<tag>.npz.  Run it twice (two libraries via DVG_LIBRARY, or two modes) and compare with
tools/fixture_tie_compare.py: which pooling windows pick another element, and how close the fixture's windows are to a tie.
    python tools/fixture_tie_probe.py f32x3 base ; python tools/fixture_tie_probe.py f32x3 new ; python tools/fixture_tie_compare.py"""
import sys, os; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np, torch, gen
from image_generation_amd import _lib, dev
from image_generation_amd.model_wrapper import ModelWrapper
ROOT = "/root/repo"
mode, tag = sys.argv[1], sys.argv[2]
_lib.set_conv_precision(mode)
params = os.path.join(ROOT, "tests", "golden", "step_params.yaml")
model = ModelWrapper("Advantage_system4", n_latents=64, training_parameter_file=params)
B = model.BATCH_SIZE
images = torch.from_numpy(gen.make_images(B * 12, seed=909)).reshape(12, B, 1, 32, 32)
model.set_dataloader([(images[k], torch.zeros(B)) for k in range(12)])
model.train_init(n_epochs=1)
enc = model._dvae.encoder.train()
lg = enc(images[0].cuda())
ws = lg.grad_fn.saved_tensors[1]
saved = dev.encoder_saved(ws, B, 64)
out = {}
bns = [m for m in enc.conv if isinstance(m, torch.nn.BatchNorm2d)]
for l in (1, 2, 3):
    y = saved[l]["Y"].cpu().numpy(); mu = saved[l]["mean"].cpu().numpy(); is_ = saved[l]["invstd"].cpu().numpy()
    g = bns[l].weight.detach().cpu().numpy(); b = bns[l].bias.detach().cpu().numpy()
    zh = ((y - mu[None, :, None, None]) * is_[None, :, None, None]).astype(np.float32)
    z = (zh.astype(np.float64) * g[None, :, None, None] + b[None, :, None, None]).astype(np.float32)
    out[f"z{l}"] = z; out[f"y{l}"] = y
np.savez(f"/tmp/tie_{tag}.npz", **out)
