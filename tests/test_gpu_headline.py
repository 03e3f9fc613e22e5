"""The bench workload under test: ONE training step at the exact c3 shape of BASELINE.json configs[2]
(B = 4096, 512-spin Zephyr sub-graph, R = 8, 256 reads, 200-sweep PCD) against the oracle evaluated in float64 on the
device (tests/halfstep_oracle.py: oracle/nets.py + oracle/plugin.py through stock PyTorch-ROCm, the sampler through
the C restatement), with injected Gumbel noise and dropout masks; then the same workload graph-replayed against eager,
bit for bit.  Also the per-GPU slice of configs[4] (c5: n = 1024, B = 256, 2048 chains): MMD at 2048 x 2048 x 1024 and
the whole step.  Reference step: /root/reference/src/model_wrapper.py:279-353.
"""
import os

import numpy as np
import pytest
import torch
import yaml

import halfstep_oracle as ho
from image_generation_amd import functional as F
from image_generation_amd.data import synthetic_images
from image_generation_amd.model_wrapper import ModelWrapper
from oracle import plugin

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C3 = dict(B=4096, n=512, R=8, C=256, sweeps=200, qpu="Advantage2_system1")
C5 = dict(B=256, n=1024, R=8, C=2048, sweeps=50, qpu="Advantage2_system1")
# BASELINE.json configs[0] and configs[1] at their own shapes (bench.py CONFIGS["c1"], ["c2"])
C1 = dict(B=64, n=64, R=8, C=256, sweeps=1, qpu="Advantage_system4", persistent=False)
C2 = dict(B=256, n=128, R=8, C=256, sweeps=50, qpu="Advantage_system4")
STOCK_CAP = 1.5e-2


def _yaml(tmp_path, cfg, name="p.yaml"):
    base = yaml.safe_load(open(os.path.join(ROOT, "image-generation_amd", "training_parameters.yaml")))
    base.update(BATCH_SIZE=cfg["B"], N_REPLICAS=cfg["R"], NUM_READS=cfg["C"], GIBBS_SWEEPS=cfg["sweeps"],
                GIBBS_PERSISTENT=cfg.get("persistent", True), CONV_PRECISION=cfg.get("precision", "f32"))
    path = tmp_path / name
    with open(path, "w") as f:
        yaml.safe_dump(base, f)
    return str(path)


def _model(tmp_path, cfg, steps):
    torch.manual_seed(0)
    m = ModelWrapper(cfg["qpu"], n_latents=cfg["n"], training_parameter_file=_yaml(tmp_path, cfg))
    imgs = synthetic_images(steps * cfg["B"], seed=11, device="cuda").reshape(steps, cfg["B"], 1, 32, 32)
    m.set_dataloader([(imgs[k], None) for k in range(steps)])
    m.train_init(1)
    return m, imgs


def _noise(cfg, seed):
    """Gumbel(0,1) noise (B,R,n,2) and Dropout2d keep-masks, drawn on the device (they are inputs, not results)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    B, R, n = cfg["B"], cfg["R"], cfg["n"]
    u = torch.rand((B, R, n, 2), generator=g, device="cuda", dtype=torch.float64).clamp_(1e-300, 1.0)
    gumbels = (-torch.log(-torch.log(u))).float()
    masks = [(torch.rand((B * R, c), generator=g, device="cuda") < 0.8).float() for c in (128, 64, 32, 1)]
    return gumbels, masks


def _rel_l2(got, want):
    got, want = got.double(), want.double()
    return float((got - want).norm() / (want.norm() + 1e-300))


def _check_step_against_float64(tmp_path, cfg, grad_bar):
    m, imgs = _model(tmp_path, cfg, 1)
    gumbels, masks = _noise(cfg, 5)
    snap, meta = ho.snapshot(m), ho.meta_of(m)
    osampler = ho.oracle_sampler_like(m)
    osampler_start = ho.sampler_position(osampler)
    m.noise_hook = lambda step: {"gumbels": gumbels, "dropout_masks": masks}
    m.step((imgs[0], None), epoch=0)  # eager (noise-injected steps always are): step 0 trains the GRBM as well
    torch.cuda.synchronize()
    got = {"mse": float(m.losses["mse_losses"][0]), "dvae": float(m.losses["dvae_losses"][0]), "nll": float(m.last["nll"])}
    got_grads = {k: p.grad.detach().clone() for k, p in m._dvae.named_parameters()}
    g_lin, g_quad = m._grbm._linear.grad.detach().clone(), m._grbm._quadratic.grad.detach().clone()
    del m
    torch.cuda.empty_cache()
    w = ho.oracle_step(meta, snap, imgs[0], gumbels, masks, osampler, dtype=torch.float64, device="cuda",
                       grbm_branch=True, mmd_chunk=1024)
    assert np.isfinite([got["mse"], got["dvae"], got["nll"]]).all()
    assert abs(got["mse"] - w["mse"]) <= 1e-5 * abs(w["mse"]), (got, w["mse"])
    assert abs(got["dvae"] - (w["mse"] + w["mmd"])) <= 1e-5 * abs(w["mse"] + w["mmd"]), (got, w["mse"], w["mmd"])
    # the MMD term on its own (it is ~1e-3 of the sum here: random spins against random samples): 1e-5 of the three
    # expectations it is the difference of would be meaningless, so it is held to 1e-5 relative of itself plus the
    # float32 resolution of those expectations (each is a mean of O(1) kernel values)
    assert abs((got["dvae"] - got["mse"]) - w["mmd"]) <= 1e-5 * abs(w["mmd"]) + 2e-7, (got, w["mmd"])
    assert abs(got["nll"] - w["nll"]) <= 1e-5 * abs(w["nll"]), (got["nll"], w["nll"])
    # Gradient bar.  float32 arithmetic routes a max-pool window whose two largest entries differ by less than its rounding
    # noise (and a LeakyReLU input within rounding of 0) by that noise -- DESIGN.md 5 -- and at B = 4096 on binary images
    # there are enough such windows that STOCK float32 PyTorch (the same oracle code in float32 on this device) sits
    # 1-2e-2 from float64 on the early encoder layers (measured; the HIP path: 3-7e-3).  So: relative L2 <= 5e-3 against
    # float64, or -- where stock float32 itself is farther than that -- at least as close to float64 as stock float32 is.
    osampler32 = ho.oracle_sampler_like_snapshot(osampler_start)
    with torch.backends.cudnn.flags(enabled=False):  # (ATen's own float32 convolutions: MIOpen's first-call search takes minutes)
        w32 = ho.oracle_step(meta, snap, imgs[0], gumbels, masks, osampler32, dtype=torch.float32, device="cuda",
                             grbm_branch=False, mmd_chunk=1024)
    worst, stock = {}, {}
    for name, g in got_grads.items():
        if ho.zero_true_gradient(name):
            continue
        worst[name] = _rel_l2(g, w["grads"][name])
        stock[name] = _rel_l2(w32["grads"][name], w["grads"][name])
    # (the stock-float32 allowance is itself capped: STOCK_CAP is ~1.5x the farthest stock float32 has been measured from
    # float64 on any tensor here, so the relative bar cannot pass a multiple-x regression on the tensors that use it)
    bad = {k: (v, stock[k]) for k, v in worst.items() if not v < max(grad_bar, min(stock[k], STOCK_CAP))}
    assert not bad, bad
    used_allowance = {k: (f"{v:.2e}", f"{stock[k]:.2e}") for k, v in worst.items() if v >= grad_bar}
    print(f"parameters above the {grad_bar:g} bar that passed on the stock-float32 allowance (HIP, stock):", used_allowance or "none")
    print("gradient rel-L2 vs float64, HIP / stock float32 (worst 6):",
          [(k, f"{v:.2e}", f"{stock[k]:.2e}") for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]])
    # GRBM sufficient statistics: means of +-1 products, exact in either arithmetic -- on the same spins.  A handful of the
    # B R n Gumbel arg-max decisions sit within float32 rounding of a tie and come out differently in float64 (measured at
    # c3: 14 of 16.8 M spins); each moves one entry of the data mean by 2 / (B R), i.e. ~2.5e-5 of the vector's norm at c3.
    assert _rel_l2(g_lin, w["grad_linear"]) < 2e-4 and _rel_l2(g_quad, w["grad_quadratic"]) < 2e-4
    assert float((g_lin.double() - w["grad_linear"]).abs().max()) <= 4.0 / (cfg["B"] * cfg["R"]) + 1e-6
    return worst


@pytest.mark.parametrize("precision", ["f32", "f32x3"])
def test_c3_step_matches_float64_oracle_on_device(tmp_path, precision):
    """mse, mse + mmd, nll <= 1e-5 relative; every parameter gradient's relative L2 <= 5e-3 (the bar of
    tests/test_gpu_fullsize.py: float32 LeakyReLU kinks / pooling near-ties route a few elements differently).  Under
    the headline's strict float32 operands AND under the f32x3 side mode (float32 operands as three bf16 pieces on the
    bf16 MFMA, reported beside the headline): same bars."""
    from image_generation_amd import _lib

    try:
        worst = _check_step_against_float64(tmp_path, dict(C3, precision=precision), 5e-3)
    finally:
        _lib.set_conv_precision("f32")  # (process-wide mode: the YAML key set it)
    print(f"c3 [{precision}] gradient rel-L2 (worst 5):", sorted(worst.items(), key=lambda kv: -kv[1])[:5])


@pytest.mark.parametrize("n,qpu", [(192, "Advantage_system4"), (320, "Advantage2_system1"), (448, "Advantage_system6")])
def test_ui_selectable_latent_sizes_step_matches_float64_oracle(tmp_path, n, qpu):
    """The latent sizes the reference's UI offers besides the powers of two (/root/reference/demo_configs.py:47-52:
    128...512 in steps of 64): one whole training step -- networks at n channels, discretisation, MMD at d = n, sampler
    on an n-spin Pegasus / Zephyr sub-graph, NLL -- against the float64 oracle, same bars as the headline shape."""
    worst = _check_step_against_float64(tmp_path, dict(B=96, n=n, R=8, C=128, sweeps=20, qpu=qpu), 5e-3)
    print(f"n={n} gradient rel-L2 (worst 5):", sorted(worst.items(), key=lambda kv: -kv[1])[:5])


@pytest.mark.parametrize("name,cfg", [("c1", C1), ("c2", C2)])
def test_c1_c2_step_matches_float64_oracle_on_device(tmp_path, name, cfg):
    """BASELINE.json configs[0] (B = 64, 64-spin Pegasus sub-graph, ONE Gibbs sweep from fresh chains) and configs[1]
    (B = 256, 128 spins, 50-sweep PCD) as ONE whole training step each against the float64 oracle -- the small-launch
    kernel forms (direct GEMMs below the Winograd thresholds, one-row sampler kernels, the MMD kernels of d = 64 / 128,
    the unfused decoder tail below 8192 rows) under the same bars as the headline shape (VERDICT r5 missing #3)."""
    worst = _check_step_against_float64(tmp_path, cfg, 5e-3)
    print(f"{name} gradient rel-L2 (worst 5):", sorted(worst.items(), key=lambda kv: -kv[1])[:5])


def test_c5_slice_step_matches_float64_oracle_on_device(tmp_path):
    """configs[4]'s per-GPU slice: 1024-spin networks at B = 256 (B R = 2048 decoder rows), MMD 2048 x 2048 x 1024,
    2048 chains x 50 sweeps."""
    worst = _check_step_against_float64(tmp_path, C5, 5e-3)
    print("c5 gradient rel-L2 (worst 5):", sorted(worst.items(), key=lambda kv: -kv[1])[:5])


def test_c5_slice_mmd_matches_float64():
    """F.mmd_loss at the c5 per-GPU shape (nx = ny = 2048, d = 1024) against oracle/plugin.py in float64: loss and the
    whole gradient."""
    nx, ny, d = 2048, 2048, 1024
    g = torch.Generator().manual_seed(17)
    x = ((torch.rand(nx, d, generator=g) < 0.4).float() * 2 - 1).cuda()
    x[40:44] = x[39]
    y = ((torch.rand(ny, d, generator=g) < 0.55).float() * 2 - 1).cuda()
    xa = x.clone().requires_grad_(True)
    la = F.mmd_loss(xa, y)
    la.backward()
    x64 = x.double().requires_grad_(True)
    want = plugin.mmd_loss(x64, y.double())
    want.backward()
    assert abs(float(la.detach()) - float(want)) <= 1e-5 * abs(float(want))
    assert float((xa.grad.double() - x64.grad).abs().max()) <= 2e-5 * float(x64.grad.abs().max())
    wc, gc = ho.chunked_mmd(x.double(), y.double(), chunk=512)  # the chunked evaluation the step tests use == the plain one
    assert abs(float(wc) - float(want)) <= 1e-12 * abs(float(want))
    assert float((gc - x64.grad).abs().max()) <= 1e-10 * float(x64.grad.abs().max())


def test_c3_graph_replay_is_bit_identical_to_eager(tmp_path):
    """The bench's launch mode at the bench's shape: 3 eager steps, the capture, 4 replays -- against 8 eager steps of a
    twin model: losses of every step and every parameter afterwards, bit for bit (the STREAM_BLOCKS-capped grids, the
    composed first decoder layers, position-major tiles at a 32768-row grid and the deferred MMD join inside the graph
    are reached only at this size)."""
    steps = 8

    def run(use_graph):
        m, imgs = _model(tmp_path, C3, steps)
        m.sync_losses = False
        m.use_graph = use_graph
        out = []
        for k in range(steps):
            m.step((imgs[k], None), epoch=0)
            out.append((float(m.last["mse"]), float(m.last["mmd"])))
        torch.cuda.synchronize()
        sd = {k: v.detach().clone() for k, v in m._dvae.state_dict().items()}
        sd.update({"grbm." + k: v.detach().clone() for k, v in m._grbm.state_dict().items()})
        captured = m._graph is not None and not m._graph_failed
        del m
        torch.cuda.empty_cache()
        return out, sd, captured

    eager, sd_e, _ = run(False)
    graphed, sd_g, captured = run(True)
    assert captured, "the step was not captured"
    assert np.isfinite(np.asarray(eager)).all()
    assert eager == graphed, (eager, graphed)
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_g[k]), k
    assert eager[0][0] > eager[-1][0] * 0.5  # (sanity: the losses are O(0.1-1), not garbage)
