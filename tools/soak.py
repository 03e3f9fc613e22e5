"""Graph-replayed training steps (default 600 at the bench shape c2; `soak.py <config> <steps> [f32|bf16]`) on structured synthetic images; prints the losses every 100 steps
and asserts that every parameter is finite (the run that exposed the u = 1 Gumbel draw)."""
import sys, torch, tempfile
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from image_generation_amd.model_wrapper import ModelWrapper
from image_generation_amd.data import synthetic_images
CFG = sys.argv[1] if len(sys.argv) > 1 else "c2"
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 600
cfg = dict(bench.CONFIGS[CFG])
tmp = tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False); tmp.close(); bench.write_yaml(cfg, tmp.name, sys.argv[3] if len(sys.argv) > 3 else "f32")
m = ModelWrapper(cfg["qpu"], n_latents=cfg["n"], training_parameter_file=tmp.name)
# structured images (blobs) so that there is something to learn
g = torch.Generator().manual_seed(0)
base = (torch.rand(64, 1, 8, 8, generator=g) < 0.3).float()
imgs = torch.nn.functional.interpolate(base, size=(32, 32), mode="nearest")
idx = torch.randint(0, 64, (STEPS, cfg["B"]), generator=g)
imgs = imgs.cuda(); idx = idx.cuda()
m.set_dataloader([(None, None)] * STEPS); m.train_init(1)
m.sync_losses = False; m.use_graph = True
for k in range(STEPS):
    m.step((imgs[idx[k]], None), 0)
    if k % max(1, STEPS // 6) == 0 or k == STEPS - 1:
        torch.cuda.synchronize()
        print(k, float(m.last["mse"]), float(m.last["mmd"]), float(m.last.get("nll", torch.tensor(float("nan")))), flush=True)
sd = m._dvae.state_dict()
assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
print("SOAK_OK")
