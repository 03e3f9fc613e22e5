"""Host-side logic that needs no GPU: the picture grid of the generation entry points, the data-parallel sharding of
the input pipeline, the reference-named ``ModelWrapper`` entry points (/root/reference/src/model_wrapper.py:355-491)."""
import inspect
import json
import os

import numpy as np
import pytest
import torch

from image_generation_amd import data, viz
from image_generation_amd.model_wrapper import ModelWrapper


def test_make_grid_layout():
    imgs = torch.arange(5 * 2 * 3, dtype=torch.float32).reshape(5, 1, 2, 3)
    g = viz.make_grid(imgs, nrow=4, padding=1, pad_value=-1.0)
    assert g.shape == (3, 2 * 3 + 1, 4 * 4 + 1)  # 2 rows x 4 columns of (2+1) x (3+1) cells, + the closing frame
    assert torch.equal(g[0], g[1]) and torch.equal(g[1], g[2])  # single channel replicated
    assert torch.equal(g[0, 1:3, 1:4], imgs[0, 0]) and torch.equal(g[0, 1:3, 13:16], imgs[3, 0])
    assert torch.equal(g[0, 4:6, 1:4], imgs[4, 0])
    assert bool((g[0, 0] == -1).all()) and bool((g[0, :, 0] == -1).all()) and bool((g[0, 4:6, 5:] == -1).all())
    g0 = viz.make_grid(imgs[:4], nrow=16, padding=0)
    assert g0.shape == (3, 2, 12) and torch.equal(g0[0, :, 3:6], imgs[1, 0])


def test_sharpen_rule():
    x = torch.tensor([0.0, 0.39, 0.4, 0.41, 0.6, 0.61, 1.0])
    assert viz.LOWER_THRESHOLD == 0.4 and viz.UPPER_THRESHOLD == 0.6  # /root/reference/demo_configs.py:62-63
    # H(0) = 0 on both thresholds: 0.4 -> 0, 0.6 stays
    np.testing.assert_allclose(viz.sharpen(x).numpy(), [0.0, 0.0, 0.0, 0.41, 0.6, 1.0, 1.0])


def test_tensor_batches_shards_are_disjoint_and_cover_one_permutation():
    imgs = torch.arange(103, dtype=torch.float32).reshape(103, 1, 1, 1)
    labels = torch.zeros(103, dtype=torch.int64)
    one = data.TensorBatches(imgs, labels, 5, seed=9)
    ranks = [data.TensorBatches(imgs, labels, 5, seed=9, rank=r, world_size=4) for r in range(4)]
    assert len(one) == 20 and all(len(r) == 5 for r in ranks)  # (103 // 4) // 5
    for _epoch in range(2):  # the permutation generator advances identically on every rank
        whole = torch.cat([b for b, _ in one]).flatten()
        seen = [torch.cat([b for b, _ in r]).flatten() for r in ranks]
        allv = torch.cat(seen)
        assert len(set(allv.tolist())) == allv.numel() == 100
        # rank r holds entries r, r + 4, ... of the same permutation the single-rank loader walks
        perm = torch.cat([whole, torch.tensor(sorted(set(range(103)) - set(whole.tolist())), dtype=torch.float32)])
        for r in range(4):
            assert set(seen[r].tolist()) <= set(perm.tolist())
    a = [b.flatten().tolist() for b, _ in data.TensorBatches(imgs, labels, 5, seed=9, rank=1, world_size=4)]
    b = [b.flatten().tolist() for b, _ in data.TensorBatches(imgs, labels, 5, seed=9, rank=1, world_size=4)]
    assert a == b


def test_reference_entry_points_exist_with_the_reference_signatures():
    sig = lambda f: list(inspect.signature(f).parameters)  # noqa: E731
    assert sig(ModelWrapper.generate_output) == ["self", "latent_qpu_file", "sharpen", "save_to_file"]
    assert sig(ModelWrapper.generate_reconstucted_samples) == ["self", "sharpen", "save_to_file"]
    assert sig(ModelWrapper.generate_loss_plot) == ["self", "save_to_file_mse", "save_to_file_total", "old_loss_data"]
    for name in ("setup", "train_init", "step", "save", "load", "_load_dataset"):
        assert callable(getattr(ModelWrapper, name))


def test_generate_loss_plot_figures_and_side_files(tmp_path, golden_dir):
    pytest.importorskip("plotly")
    m = ModelWrapper("Advantage_system4", n_latents=64, training_parameter_file=os.path.join(golden_dir, "step_params.yaml"))
    m.losses = {"mse_losses": [0.3, torch.tensor(0.2)], "dvae_losses": [0.5, torch.tensor(0.4)]}
    old = {"mse_losses": [0.9], "dvae_losses": [1.1]}
    f_mse, f_tot = m.generate_loss_plot(str(tmp_path / "mse.json"), str(tmp_path / "tot.json"), old_loss_data=old)
    np.testing.assert_allclose(list(f_mse.data[0].y), [0.9, 0.3, 0.2], rtol=1e-6)
    np.testing.assert_allclose(list(f_tot.data[0].y), [1.1, 0.5, 0.4], rtol=1e-6)
    assert list(f_tot.data[0].x) == [0, 1, 2] and f_mse.layout.xaxis.title.text == "Batch" and f_mse.layout.yaxis.title.text == "Loss"
    d = json.load(open(tmp_path / "tot.json"))
    assert d["data"][0]["type"] == "scatter" and d["layout"]["margin"] == {"t": 0, "l": 0, "b": 0, "r": 0}


def test_dataset_size_is_the_references_random_split_subset():
    """/root/reference/src/model_wrapper.py:96-100: ``random_split(dataset, [k, n - k])[0]`` -- a random subset drawn from
    torch's global generator, not the first k images; data-parallel ranks (seeded form) agree among themselves."""
    torch.manual_seed(5)
    want = torch.utils.data.random_split(range(1000), [100, 900])[0].indices
    torch.manual_seed(5)
    got = data.random_subset_indices(1000, 100)
    assert got.tolist() == want and sorted(want) != list(range(100)) and len(set(want)) == 100
    a, b = data.random_subset_indices(1000, 100, seed=3), data.random_subset_indices(1000, 100, seed=3)
    assert torch.equal(a, b) and not torch.equal(a, got)
    with pytest.raises(ValueError):
        data.random_subset_indices(10, 11)
    torch.manual_seed(5)
    dl = data.get_dataloader(32, 10, dataset_size=100, seed=1, device="cpu")
    assert len(dl) == 10 and dl.images.shape == (100, 1, 32, 32)


def test_schedule_decisions_of_the_wrapper_need_no_gpu():
    """Round 4's host-side schedule decisions: the decoder's weight-only prologue is forked only from
    ``PREPARE_DECODER_ROWS`` decoder rows up (a fork costs a captured step more than the prologue takes below that) unless
    ``prepare_decoder`` forces it; on a CPU device both the preparation and the device-hold of the join measurement are
    no-ops; the MMD-join decision takes the LARGEST measured lag (a host-bound sample reads ~0)."""
    from image_generation_amd.modules import Decoder

    m = ModelWrapper.__new__(ModelWrapper)
    m.prepare_decoder, m._prep_stream, m._device, m.N_REPLICAS = None, None, torch.device("cpu"), 8
    assert ModelWrapper.PREPARE_DECODER_ROWS == 32768
    m._prepare_decoder(torch.zeros(4096, 1, 32, 32), None)   # 32768 rows, but no GPU: nothing happens
    assert m._prep_stream is None
    m.n_latents = 1024                                        # (round 6: ... and from 512 latent spins up, rows x spins >= 2^20)
    m._prepare_decoder(torch.zeros(256, 1, 32, 32), None)
    assert m._prep_stream is None
    m.overlap_sampler = m.overlap_mmd = True
    m.sampler = None
    m._hold_device_while_measuring(4096 * 8)                  # (returns before touching torch.cuda)
    dec = Decoder(32)
    dec.prepare(64, None)                                     # CPU parameters: a no-op, nothing pending
    assert dec._prepared is None
    # the decision rule itself
    m.DEFER_SAMPLES, m.DEFER_LAG_MS = 2, 0.05
    rec = {"decision": None, "events": [], "lags": [0.004, 0.9]}
    m.__dict__["_defer_state"] = {((16, 32), (4, 32), 0): rec}

    class _Ev:  # two already-synchronised "event pairs" whose lags are 0.004 and 0.9 ms
        def __init__(self, v): self.v = v
        def synchronize(self): pass
        def elapsed_time(self, other): return other.v - self.v
    rec["lags"], rec["events"] = [], [(_Ev(0.0), _Ev(0.004)), (_Ev(0.0), _Ev(0.9))]
    m.__dict__["_defer_rec"] = rec
    assert m._defer_undecided() is True                       # (a shape still being measured is not captured into a graph)
    assert m._defer_mmd_join(torch.empty(16, 32), torch.empty(4, 32)) is True and rec["lags"] == [0.004, 0.9]
    assert m._defer_undecided() is False and m.__dict__["_defer_by_rows"][16] is rec
