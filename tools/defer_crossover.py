#!/usr/bin/env python3
"""Where does joining the MMD stream BEHIND the decoder's backward (ModelWrapper._defer_mmd_join) start to pay?

    python tools/defer_crossover.py            # on an MI355X; prints one line per shape

For every shape the graph-replayed training step is timed with the join forced in front of the decoder's backward
(defer = 0) and behind it (defer = 1); `work` is the predicate's argument, (B R + C) * n.  The rule in model_wrapper.py
is the measured crossover of this table (profiles/r04_defer_mmd_join_crossover.txt)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import yaml  # noqa: E402

from image_generation_amd.data import synthetic_images  # noqa: E402
from image_generation_amd.model_wrapper import ModelWrapper  # noqa: E402


def step_ms(n, B, C, sweeps, qpu, defer, steps=40):
    base = yaml.safe_load(open(os.path.join(ROOT, "image-generation_amd", "training_parameters.yaml")))
    base.update(BATCH_SIZE=B, N_REPLICAS=8, NUM_READS=C, GIBBS_SWEEPS=sweeps, GIBBS_PERSISTENT=True, CONV_PRECISION="f32")
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(base, f)
    torch.manual_seed(0)
    m = ModelWrapper(qpu, n_latents=n, training_parameter_file=f.name)
    os.unlink(f.name)
    imgs = synthetic_images(8 * B, seed=3, device="cuda").reshape(8, B, 1, 32, 32)
    m.set_dataloader([(imgs[k % 8], None) for k in range(steps + 40)])
    m.train_init(1)
    m.sync_losses, m.keep_step_losses, m.use_graph = False, False, True
    m.defer_mmd_join = bool(defer)
    k = 0
    for _ in range(12):
        m.step((imgs[k % 8], None), epoch=0); k += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.step((imgs[k % 8], None), epoch=0); k += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


shapes = [(128, 256, 256, 50, "Advantage_system4"), (256, 256, 256, 50, "Advantage_system4"), (512, 128, 256, 50, "Advantage2_system1"),
          (256, 1024, 256, 50, "Advantage_system4"), (512, 512, 256, 50, "Advantage2_system1"), (512, 1024, 256, 100, "Advantage2_system1"),
          (1024, 256, 2048, 50, "Advantage2_system1"), (512, 2048, 256, 200, "Advantage2_system1"), (512, 4096, 256, 200, "Advantage2_system1")]
print(f"{'n':>5s} {'B':>5s} {'C':>5s} {'work (B R + C) n':>17s} {'join first ms':>13s} {'join behind ms':>14s} {'behind / first':>14s}")
for n, B, C, sweeps, qpu in shapes:
    a = min(step_ms(n, B, C, sweeps, qpu, 0) for _ in range(2))
    b = min(step_ms(n, B, C, sweeps, qpu, 1) for _ in range(2))
    print(f"{n:5d} {B:5d} {C:5d} {(B * 8 + C) * n:17d} {a:13.4f} {b:14.4f} {b / a:14.3f}", flush=True)
