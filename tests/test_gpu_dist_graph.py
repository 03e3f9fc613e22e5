"""Data-parallel graph replay (two graphs per step + the eager all-reduce between them) on ONE GPU: the collective path
is forced with a single-rank RCCL group (DVG_FORCE_DIST=1) in a child process, and the replayed trajectory must be
bit-identical to the eager data-parallel one."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.environ["DVG_REPO"])
sys.path.insert(0, os.path.join(os.environ["DVG_REPO"], "tests", "golden"))
import gen
from image_generation_amd.model_wrapper import ModelWrapper
from image_generation_amd.parallel import DataParallel

dp = DataParallel()
assert dp.force and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
params = os.path.join(os.environ["DVG_REPO"], "tests", "golden", "step_params.yaml")

def run(use_graph):
    torch.manual_seed(0)
    m = ModelWrapper("Advantage_system4", n_latents=64, training_parameter_file=params, dist=dp)
    B = m.BATCH_SIZE
    imgs = torch.from_numpy(gen.make_images(B * 14, seed=4)).reshape(14, B, 1, 32, 32).cuda()
    m.set_dataloader([(imgs[k], None) for k in range(14)])
    m.train_init(1)
    m.sync_losses = False
    m.use_graph = use_graph
    out = []
    for k in range(14):
        m.step((imgs[k], None), epoch=0)
        out.append((float(m.last["mse"]), float(m.last["mmd"])))
    torch.cuda.synchronize()
    sd = {k: v.clone() for k, v in m._dvae.state_dict().items()}
    sd.update({"grbm." + k: v.clone() for k, v in m._grbm.state_dict().items()})
    return out, sd, m

eager, sd_e, _ = run(False)
graphed, sd_g, mg = run(True)
assert mg._graphs and mg._graphs[0][3] is not None and not mg._graph_failed, "the split capture did not happen"
assert eager == graphed, (eager, graphed)
for k in sd_e:
    assert torch.equal(sd_e[k], sd_g[k]), k
dp.shutdown()
print("DIST_GRAPH_OK")
'''


def test_data_parallel_graph_replay_matches_eager():
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DVG_FORCE_DIST="1", DVG_REPO=repo, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_GRAPH_OK" in r.stdout, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
