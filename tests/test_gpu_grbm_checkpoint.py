"""The GRBM half of SHIPPED checkpoints on the GPU (VERDICT r3 missing #2): trained ``h, J`` (|J| mean 2.4, max 4.9) on
real-QPU sub-graphs -- Zephyr ``models/Advantage2_system1_40_epochs/grbm.pth`` (2059 edges) and Pegasus
``models/Advantage_system6_10_epochs/grbm.pth`` (1635 edges) -- in the checkpoint's OWN edge order, committed as data in
tests/golden/grbm_ckpt.npz with (a) oracle draws and float64 energies and (b) what the reference's VERBATIM
``ModelWrapper.load`` + ``generate_output`` (/root/reference/src/model_wrapper.py:164-175,355-399; the call pair of
demo_callbacks.py:757-758) produced from them over the oracle sampler (tests/golden/make_golden.py ``grbm_checkpoint``).

Checked here: the HIP sampler is bit-exact on the checkpoint graphs (fresh draw, persistent second draw, and a draw with
``to_ising``'s clamp binding), the energy kernel is within 1e-6, and ``ModelWrapper.load`` of the checkpoint folder
followed by ``generate_output`` reproduces the reference's pictures and latent file."""
import json
import os

import numpy as np
import pytest
import torch

import gen
from image_generation_amd import graphs, sampler as smp
from image_generation_amd.plugin import GraphRestrictedBoltzmannMachine
from oracle import cref, gibbs


@pytest.fixture(scope="module")
def fx(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "grbm_ckpt.npz")))


def _plan(fx, fam):
    ei, ej = fx[f"{fam}/edge_i"].astype(np.int64), fx[f"{fam}/edge_j"].astype(np.int64)
    return graphs.build_plan(256, ei, ej), ei, ej


@pytest.mark.parametrize("fam,n_edges", [("zephyr", 2059), ("pegasus", 1635)])
def test_fixture_is_a_trained_grbm_and_the_oracle_reproduces_its_draws(fx, fam, n_edges):
    """(CPU) the data are a trained GRBM on a real-QPU sub-graph (edge counts, i < j grouped by i,
    non-bipartite, |J| mean about 2.4), and the C restatement of the sampler reproduces the committed draws on it."""
    plan, ei, ej = _plan(fx, fam)
    J, h = fx[f"{fam}/quadratic"], fx[f"{fam}/linear"]
    assert plan.n_edges == n_edges == J.size and (ei < ej).all() and (np.diff(ei) >= 0).all()
    assert 2.0 < float(np.abs(J).mean()) < 2.8 and float(np.abs(J).max()) > 4.5 and float(np.abs(h).mean()) < 0.1
    assert plan.n_colours >= 3  # triangles: no checkerboard schedule on these graphs
    seed, sweeps, reads, pre = int(fx["seed"]), int(fx["sweeps"]), int(fx["reads"]), float(fx["prefactor"])
    hr, jr = tuple(fx[f"{fam}/h_range"]), tuple(fx[f"{fam}/j_range"])
    hs, Js = gibbs.scaled_fields(h, J, pre, hr, jr)
    ids = np.arange(reads, dtype=np.uint32)
    tail = (plan.order, plan.class_ptr, plan.adj_ptr, plan.adj_idx, plan.adj_eid, seed)
    d1 = cref.gibbs_sweeps(cref.init_state(ids, 256, seed), ids, hs, Js, 1.0 / pre, *tail, 0, sweeps)
    assert np.array_equal(d1, fx[f"{fam}/draw1"])
    d2 = cref.gibbs_sweeps(d1.copy(), ids, hs, Js, 1.0 / pre, *tail, sweeps, sweeps)
    assert np.array_equal(d2, fx[f"{fam}/draw2"])
    x = d2.astype(np.float64)
    np.testing.assert_allclose(x @ h.astype(np.float64) + (x[:, ei] * x[:, ej]) @ J.astype(np.float64), fx[f"{fam}/energy"], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("fam", ["zephyr", "pegasus"])
def test_sampler_bit_exact_and_energy_on_checkpoint_graph(fx, fam):
    plan, ei, ej = _plan(fx, fam)
    seed, sweeps, reads, pre = int(fx["seed"]), int(fx["sweeps"]), int(fx["reads"]), float(fx["prefactor"])
    hr, jr = tuple(float(v) for v in fx[f"{fam}/h_range"]), tuple(float(v) for v in fx[f"{fam}/j_range"])
    lin = torch.from_numpy(fx[f"{fam}/linear"]).cuda()
    quad = torch.from_numpy(fx[f"{fam}/quadratic"]).cuda()
    s = smp.GibbsSampler(plan, list(range(256)), beta=1.0 / pre, sweeps=sweeps, seed=seed, persistent=True,
                         h_range=hr, j_range=jr)
    for key in ("draw1", "draw2"):  # the second draw continues the first one's chains
        got = s.sample_native(lin, quad, pre, hr, jr, num_reads=reads).cpu().numpy()
        assert int((got != fx[f"{fam}/{key}"].astype(np.float32)).sum()) == 0, key
    # trained couplings beyond the solver's ranges: the clamp of to_ising is what the chains see
    s2 = smp.GibbsSampler(plan, list(range(256)), beta=2.0, sweeps=sweeps, seed=seed, persistent=False, h_range=hr, j_range=jr)
    got = s2.sample_native(lin, quad, 0.5, hr, jr, num_reads=reads).cpu().numpy()
    assert int((got != fx[f"{fam}/draw_clamped"].astype(np.float32)).sum()) == 0
    # energies of the drawn states through the plugin module (dvg_grbm_energy), checkpoint loaded strictly
    grbm = GraphRestrictedBoltzmannMachine(range(256), zip(ei.tolist(), ej.tolist()))
    sd = {"_linear": lin.cpu(), "_quadratic": quad.cpu(), "_edge_idx_i": torch.from_numpy(ei), "_edge_idx_j": torch.from_numpy(ej),
          "_visible_idx": torch.arange(256), "_hidden_idx": torch.zeros(0, dtype=torch.int64),
          "_flat_adj": torch.zeros(0, dtype=torch.int64), "_flat_j_idx": torch.zeros(0, dtype=torch.int64),
          "_bin_idx": torch.zeros(0, dtype=torch.int64)}
    grbm.load_state_dict(sd, strict=True)
    grbm = grbm.cuda()
    with torch.no_grad():
        e = grbm(torch.from_numpy(fx[f"{fam}/draw2"].astype(np.float32)).cuda()).cpu().numpy().astype(np.float64)
    want = fx[f"{fam}/energy"]
    assert float(np.abs(e - want).max()) <= 1e-6 * float(np.abs(want).max())


@pytest.mark.gpu
def test_load_then_generate_output_matches_the_reference_on_a_shipped_checkpoint(fx, golden_dir, tmp_path):
    """``ModelWrapper.load(models/Advantage2_system1_40_epochs)`` + ``generate_output`` twice (plain, sharpened)."""
    pytest.importorskip("plotly")
    from image_generation_amd.model_wrapper import ModelWrapper

    ck = dict(np.load(os.path.join(golden_dir, "ckpt_adv2_40.npz")))
    folder = tmp_path / "Advantage2_system1_40_epochs"
    folder.mkdir()
    torch.save({k[3:]: torch.from_numpy(np.array(v)) for k, v in ck.items() if k.startswith("sd/")}, folder / "dvae.pth")
    ei, ej = fx["zephyr/edge_i"].astype(np.int64), fx["zephyr/edge_j"].astype(np.int64)
    empty = torch.zeros(0, dtype=torch.int64)
    torch.save({"_linear": torch.from_numpy(fx["zephyr/linear"]), "_quadratic": torch.from_numpy(fx["zephyr/quadratic"]),
                "_edge_idx_i": torch.from_numpy(ei), "_edge_idx_j": torch.from_numpy(ej), "_visible_idx": torch.arange(256),
                "_hidden_idx": empty, "_flat_adj": empty, "_flat_j_idx": empty, "_bin_idx": empty}, folder / "grbm.pth")
    params = tmp_path / "params.yaml"
    text = open(os.path.join(golden_dir, "step_params.yaml")).read()
    text = text.replace("NUM_READS: 16", f"NUM_READS: {int(fx['reads'])}").replace("GIBBS_SWEEPS: 3", f"GIBBS_SWEEPS: {int(fx['sweeps'])}")
    params.write_text(text)
    m = ModelWrapper("Advantage2_system1", n_latents=256, training_parameter_file=str(params))
    m.set_dataloader([(torch.zeros(8, 1, 32, 32), torch.zeros(8))])
    m.load(folder)
    assert m.sampler.plan.n_edges == 2059  # rebuilt on the checkpoint's own edge list, not the ideal Zephyr sub-graph
    assert torch.equal(m._grbm._edge_idx_i.cpu(), torch.from_numpy(ei))
    latent_file = str(tmp_path / "latent.json")
    for k, sharpen in enumerate((False, True)):
        fig = m.generate_output(latent_qpu_file=latent_file, sharpen=sharpen)
        latent = np.asarray(json.load(open(latent_file)), dtype=np.float32)
        assert np.array_equal(latent, fx[f"gen{k}/latent"])  # bit-exact sampler on the trained model
        assert np.array_equal(m.sampler._state.cpu().numpy().astype(np.int8), fx[f"gen{k}/samples"])
        got, want = gen.figure_image(fig).astype(np.int32), fx[f"gen{k}/image"].astype(np.int32)
        assert got.shape == want.shape
        diff = np.abs(got - want)
        # 8-bit pictures of float32 decoders (CPU oneDNN there, MFMA here): off by one level on a few pixels at most; the
        # sharpened picture may flip a pixel that sits on a threshold
        assert (diff > 1).mean() <= (2e-4 if sharpen else 0.0) and (diff > 0).mean() < 2e-3, (k, int(diff.max()), float((diff > 0).mean()))
