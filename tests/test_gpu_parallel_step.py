"""Step-level data parallelism with TWO ranks: two processes share GPU 0 and exchange through a gloo
group (the collective is staged through the host: parallel.DataParallel._host_staged), each running real
``ModelWrapper.step``s on its own shard with its own globally numbered chains.

Checked: (1) rank r's first-step losses equal the single-rank run on shard r bit for bit (same parameters, same
rank-keyed noise); (2) a GRBM step issues ONE all-reduce (encoder/decoder gradients and the GRBM sufficient-statistic
differences in one buffer); (3) after every step the replicas are bit-identical; (4) the first update equals the Adam
update of the MEAN of the two shards' gradients; (5) graph replay (two graphs + the collective between them) is
bit-identical to the eager data-parallel run; (6) replicas that were built from different seeds are equalised by the
broadcast in ``setup``; (7) rank r's losses equal the CPU oracle on shard r and the exchanged gradient the sum of
the oracle's per-shard gradients."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, numpy as np, torch
repo = os.environ["DVG_REPO"]
sys.path.insert(0, repo); sys.path.insert(0, os.path.join(repo, "tests", "golden")); sys.path.insert(0, os.path.join(repo, "tests"))
import gen
import torch.distributed as tdist
from image_generation_amd.model_wrapper import ModelWrapper
from image_generation_amd.parallel import DataParallel

mode, out_dir, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
params = os.path.join(repo, "tests", "golden", "step_params.yaml")
torch.cuda.set_device(0)

class Shard:  # a single-rank stand-in that carries a rank (seeds, chain numbering) but no process group
    def __init__(self, rank): self.rank, self.world_size, self.force, self.local_rank = rank, 1, False, 0

def images(rank, B):
    return torch.from_numpy(gen.make_images(B * steps, seed=40 + rank)).reshape(steps, B, 1, 32, 32).cuda()

def run(dist, rank, use_graph=False, seed=None, oracle=False):
    m = ModelWrapper("Advantage_system4", n_latents=64, training_parameter_file=params, dist=dist)
    imgs = images(rank, 8)
    m.set_dataloader([(imgs[k], None) for k in range(steps)])
    if seed is not None:  # build the replica BEFORE train_init seeds the generator (the reference's load() path does)
        torch.manual_seed(seed)
        m.setup()
    m.train_init(1)
    m.sync_losses = False
    m.use_graph = use_graph
    rec = {"mse": [], "mmd": [], "flat": [], "gflat": []}
    if oracle:
        # SURVEY.md 8e: "rank r's losses equal the CPU oracle on shard r with the same noise; gradients equal the mean over
        # shards" -- the CPU oracle (float32, stock torch + the C sampler) steps from this rank's snapshot on this rank's
        # shard with this rank's injected Gumbel noise / dropout masks and this rank's globally numbered chains
        import halfstep_oracle as ho
        g = torch.Generator().manual_seed(900 + rank)
        B, R, n = 8, int(m.N_REPLICAS), 64
        gumbels = -torch.log(torch.empty(B, R, n, 2).exponential_(generator=g))
        masks = [(torch.rand(B * R, c, generator=g) < 0.8).float() for c in (128, 64, 32, 1)]
        m.noise_hook = lambda step: {"gumbels": gumbels, "dropout_masks": masks} if step == 0 else {}
        w = ho.oracle_step(ho.meta_of(m), ho.snapshot(m), imgs[0].cpu(), gumbels, masks, ho.oracle_sampler_like(m),
                           dtype=torch.float32, device="cpu", grbm_branch=True)
        names = [k for k, _ in m._dvae.named_parameters()]
        rec["oracle_losses"] = np.array([w["mse"], w["mmd"], w["nll"]])
        rec["oracle_grad"] = np.concatenate([w["grads"][k].reshape(-1).numpy() for k in names]
                                            + [w["grad_linear"].numpy(), w["grad_quadratic"].numpy()])
    rec["p0"] = np.concatenate([m._dvae_optimizer.flat.detach().cpu().numpy(), m._grbm_optimizer.flat.detach().cpu().numpy()])
    calls = {"n": 0}
    if dist is not None and getattr(dist, "active", False):
        orig = dist.all_reduce_sum
        def counted(t):
            calls["n"] += 1
            return orig(t)
        dist.all_reduce_sum = counted
    per_step_calls = []
    for k in range(steps):
        before = calls["n"]
        m.step((imgs[k], None), epoch=0)
        per_step_calls.append(calls["n"] - before)
        rec["mse"].append(float(m.last["mse"])); rec["mmd"].append(float(m.last["mmd"]))
        rec["flat"].append(m._dvae_optimizer.flat.detach().cpu().numpy().copy())
        rec["gflat"].append(m._grbm_optimizer.flat.detach().cpu().numpy().copy())
        if k == 0:
            rec["grad0"] = m._joint_grad.detach().cpu().numpy().copy()  # (dist: the all-reduced sum; single: the local gradient)
            rec["nll0"] = float(m.last["nll"])
    torch.cuda.synchronize()
    rec["calls"] = per_step_calls
    rec["bn"] = np.concatenate([b.detach().float().cpu().numpy().ravel() for b in m._bn_buffers()])
    rec["graph"] = int(m._graph is not None and not m._graph_failed)
    return rec, m

if mode == "single":          # one process: the two shards one after the other, no process group
    for r in range(2):
        rec, m = run(Shard(r), r)
        np.savez(os.path.join(out_dir, f"single{r}.npz"), **{k: np.asarray(v) for k, v in rec.items()})
else:
    dp = DataParallel(backend="gloo", device=torch.device("cuda", 0))
    assert dp.world_size == 2 and tdist.get_backend() == "gloo"
    rec, m = run(dp, dp.rank, use_graph=(mode == "graph"), seed=(100 + dp.rank if mode == "seeds" else None),
                 oracle=(mode == "oracle"))
    np.savez(os.path.join(out_dir, f"{mode}{dp.rank}.npz"), **{k: np.asarray(v) for k, v in rec.items()})
    dp.barrier()
    dp.shutdown()
print("WORKER_OK")
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(mode, out_dir, steps, world):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, DVG_REPO=repo, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("DVG_FORCE_DIST", None)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, mode, str(out_dir), str(steps)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0 and "WORKER_OK" in out, out[-2000:] + "\n" + err[-4000:]


def _load(out_dir, name):
    return dict(np.load(os.path.join(out_dir, name)))


def test_two_rank_steps_match_shards_and_mean_gradient_update(tmp_path):
    steps = 12  # steps 0 and 10 train the GRBM
    _launch("single", tmp_path, steps, 1)
    _launch("eager", tmp_path, steps, 2)
    s = [_load(tmp_path, f"single{r}.npz") for r in range(2)]
    d = [_load(tmp_path, f"eager{r}.npz") for r in range(2)]
    # (1) first-step losses of rank r = the single-rank run on shard r, bit for bit
    for r in range(2):
        assert d[r]["mse"][0] == s[r]["mse"][0] and d[r]["mmd"][0] == s[r]["mmd"][0] and d[r]["nll0"] == s[r]["nll0"], r
    assert s[0]["mse"][0] != s[1]["mse"][0]  # the shards do differ
    # (2) ONE collective per step, GRBM steps included
    assert list(d[0]["calls"]) == [1] * steps and list(d[1]["calls"]) == [1] * steps
    # the all-reduced buffer of step 0 is the sum of the two local gradient buffers (encoder/decoder | GRBM)
    np.testing.assert_allclose(d[0]["grad0"], s[0]["grad0"] + s[1]["grad0"], rtol=1e-6, atol=1e-9)
    assert np.array_equal(d[0]["grad0"], d[1]["grad0"])
    # (3) replicas bit-identical after every step
    for k in range(steps):
        assert np.array_equal(d[0]["flat"][k], d[1]["flat"][k]) and np.array_equal(d[0]["gflat"][k], d[1]["gflat"][k]), k
    # (4) the first update is Adam's on the MEAN gradient (coupled weight decay; at t = 1: p - lr g' / (|g'| + eps))
    nd = d[0]["flat"].shape[1]
    assert np.array_equal(s[0]["p0"], s[1]["p0"]) and np.array_equal(d[0]["p0"], s[0]["p0"])  # same seeded start everywhere
    p0_all = s[0]["p0"].astype(np.float64)
    for part, lr, wd, after in ((slice(0, nd), 1e-4, 0.01, d[0]["flat"][0]), (slice(nd, None), 1e-3, 0.01, d[0]["gflat"][0])):
        g_mean = (s[0]["grad0"][part].astype(np.float64) + s[1]["grad0"][part]) / 2
        gp = g_mean + wd * p0_all[part]
        want = p0_all[part] - lr * gp / (np.abs(gp) + 1e-8)
        # entries whose mean gradient is ~0 take their sign from rounding: compare where |g'| is resolved
        ok = np.abs(gp) > 1e-5 * np.abs(gp).max()
        assert ok.mean() > 0.9
        np.testing.assert_allclose(after[ok], want[ok], rtol=3e-7, atol=3e-7)
        # ... and differs from a single shard's own update (the exchange really happened)
        assert not np.array_equal(after, s[0]["flat"][0] if part.start == 0 else s[0]["gflat"][0])
    # BatchNorm running statistics stay per rank during training (DDP semantics) ...
    assert not np.array_equal(d[0]["bn"], d[1]["bn"])


def test_two_rank_graph_replay_is_bit_identical_to_eager(tmp_path):
    steps = 14
    _launch("eager", tmp_path, steps, 2)
    _launch("graph", tmp_path, steps, 2)
    for r in range(2):
        e, g = _load(tmp_path, f"eager{r}.npz"), _load(tmp_path, f"graph{r}.npz")
        assert int(g["graph"]) == 1, "the split capture did not happen"
        assert np.array_equal(e["mse"], g["mse"]) and np.array_equal(e["mmd"], g["mmd"])
        assert np.array_equal(e["flat"][-1], g["flat"][-1]) and np.array_equal(e["gflat"][-1], g["gflat"][-1])
        assert list(g["calls"]) == [1] * steps


def test_replicas_built_from_different_seeds_are_equalised(tmp_path):
    _launch("seeds", tmp_path, 3, 2)
    a, b = _load(tmp_path, "seeds0.npz"), _load(tmp_path, "seeds1.npz")
    for k in range(3):
        assert np.array_equal(a["flat"][k], b["flat"][k]) and np.array_equal(a["gflat"][k], b["gflat"][k])


def test_two_rank_losses_equal_cpu_oracle_on_each_shard(tmp_path):
    """SURVEY.md 8e's parity statement, literally: rank r's step-0 losses == the CPU oracle on shard r (1e-5 relative,
    identical injected noise, the rank's own chains), and the ONE all-reduced gradient buffer == the SUM over ranks of the
    oracle's gradients (the 1/world_size of the mean is Adam's grad_scale)."""
    _launch("oracle", tmp_path, 1, 2)
    d = [_load(tmp_path, f"oracle{r}.npz") for r in range(2)]
    for r in range(2):
        got = np.array([d[r]["mse"][0], d[r]["mse"][0] + d[r]["mmd"][0], d[r]["nll0"]])
        o = d[r]["oracle_losses"]
        np.testing.assert_allclose(got, [o[0], o[0] + o[1], o[2]], rtol=1e-5, err_msg=f"rank {r}")  # mse, mse + mmd, nll
        assert abs(d[r]["mmd"][0] - o[1]) <= 1e-4 * abs(o[1]), r
    assert d[0]["oracle_losses"][0] != d[1]["oracle_losses"][0]
    want = d[0]["oracle_grad"].astype(np.float64) + d[1]["oracle_grad"]
    got = d[0]["grad0"].astype(np.float64)
    assert np.array_equal(d[0]["grad0"], d[1]["grad0"])
    assert np.linalg.norm(got - want) <= 2e-4 * np.linalg.norm(want)
