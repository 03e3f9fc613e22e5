"""The dwave.* / dimod import shims (image-generation_amd/shims): the names the reference imports resolve to this
package, and -- when the reference checkout is present (build container only; never on the GPU box) -- the
reference's OWN sub-graph / sampler construction (src/utils/common.py:103-140) runs against the local solver and
lands on the same graph as this package's restatement."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = os.path.join(ROOT, "image-generation_amd", "shims")
REF = "/root/reference"


@pytest.fixture()
def shim_path():
    saved_path, saved_mods = list(sys.path), set(sys.modules)
    sys.path[:0] = [ROOT, SHIMS]
    yield
    sys.path[:] = saved_path
    for name in set(sys.modules) - saved_mods:
        if name.split(".")[0] in ("dwave", "dimod", "src", "torchvision"):
            del sys.modules[name]


def test_names_resolve(shim_path):
    from dwave.system import DWaveSampler, FixedEmbeddingComposite
    from dwave.plugins.torch.models import DiscreteVariationalAutoencoder, GraphRestrictedBoltzmannMachine
    from dwave.plugins.torch.nn.functional import maximum_mean_discrepancy_loss
    from dwave.plugins.torch.nn.modules.kernels import GaussianKernel
    from dwave.cloud import Client
    from dimod import Sampler, SampleSet, as_samples
    import image_generation_amd.plugin as plugin

    assert DiscreteVariationalAutoencoder is plugin.DiscreteVariationalAutoencoder
    assert GraphRestrictedBoltzmannMachine is plugin.GraphRestrictedBoltzmannMachine
    assert maximum_mean_discrepancy_loss is plugin.maximum_mean_discrepancy_loss and GaussianKernel is plugin.GaussianKernel
    assert Sampler is not None
    for name, kind, nodes in [("Advantage_system4", "pegasus", 5640), ("Advantage2_system1", "zephyr", 4800)]:
        q = DWaveSampler(solver=name)
        assert q.properties["topology"]["type"] == kind and len(q.to_networkx_graph()) == nodes
        assert len(q.properties["h_range"]) == 2 and len(q.properties["j_range"]) == 2 and q.solver.name == name
    assert "Advantage_system4" in [s.name for s in Client.from_config(client="qpu").get_solvers()]
    with pytest.raises(ValueError):
        DWaveSampler(solver="no_such_solver")
    with pytest.raises(ValueError):
        FixedEmbeddingComposite(DWaveSampler(solver="Advantage_system4"), {0: [30, 31]})
    ss = SampleSet.from_samples(as_samples(np.array([[1, -1, 1], [-1, -1, 1]])), vartype="SPIN", energy=np.zeros(2))
    assert ss.record.sample.shape == (2, 3) and ss.variables == [0, 1, 2] and ss.vartype == "SPIN" and len(ss) == 2


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference checkout not present")
def test_reference_sampler_construction_runs_on_the_local_solver(shim_path):
    import networkx as nx

    if "torchvision" not in sys.modules:  # imported by src/model_wrapper.py only for the dataset; absent in this image
        for mod in ("torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.utils"):
            sys.modules[mod] = types.ModuleType(mod)
    sys.path.insert(0, REF)
    from src.utils import common as ref_common  # the reference's own file, unmodified
    from image_generation_amd import graphs
    from image_generation_amd.sampler import get_sampler_and_sampler_kwargs

    sampler, kwargs, mapped, lin_r, quad_r = ref_common.get_sampler_and_sampler_kwargs(
        num_reads=16, annealing_time=1.0, n_latents=64, random_seed=1234, qpu="Advantage_system4")
    ours, okwargs, omapped, olin, oquad = get_sampler_and_sampler_kwargs(16, 1.0, 64, 1234, "Advantage_system4", device="cpu")
    assert type(sampler).__name__ == "FixedEmbeddingComposite" and kwargs == okwargs
    assert tuple(lin_r) == tuple(olin) and tuple(quad_r) == tuple(oquad)
    assert sorted(mapped.nodes) == sorted(omapped.nodes) == list(range(64))
    assert sorted(tuple(sorted(e)) for e in mapped.edges) == sorted(tuple(sorted(e)) for e in omapped.edges)
    assert nx.is_connected(mapped)
    # the composite's plan is the plan this package builds for the same sub-graph
    built = sampler._build()
    assert built.plan.n == 64 and built.plan.n_edges == ours.plan.n_edges
    pairs = lambda p: sorted(zip(np.asarray(p.edge_i).tolist(), np.asarray(p.edge_j).tolist()))  # noqa: E731
    assert pairs(built.plan) == pairs(ours.plan)  # (edge ORDER is a per-sampler detail: couplers are looked up by key)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference checkout not present")
def test_reference_generate_model_fig_runs_on_the_layout_shim(shim_path, tmp_path, monkeypatch):
    """The reference's own ``generate_model_fig`` (/root/reference/src/utils/callback_helpers.py:344-381: local solver ->
    sub-graph -> ``dnx.drawing.pegasus_layout`` / ``zephyr_layout`` -> two plotly figures) over the ``dwave_networkx``
    shim: every sub-graph node gets a coordinate, couplers join nearby points."""
    pytest.importorskip("plotly")
    for mod in ("torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.utils"):
        if mod not in sys.modules:
            sys.modules[mod] = types.ModuleType(mod)
    tv = sys.modules
    tv["torchvision.datasets"].MNIST = object
    for name in ("Compose", "Resize", "ToTensor"):
        setattr(tv["torchvision.transforms"], name, object)
    tv["torchvision.utils"].make_grid = tv["torchvision.utils"].save_image = lambda *a, **k: None
    sys.path.insert(0, REF)
    monkeypatch.chdir(tmp_path)
    os.makedirs("assets/model_diagram")
    from src.utils import callback_helpers as ref_ch  # the reference's own file, unmodified

    for qpu, n in (("Advantage_system4", 64), ("Advantage2_system1", 128)):
        fig_qpu, fig_not_qpu, latent_mapping = ref_ch.generate_model_fig(qpu, n, 1234)
        assert sorted(latent_mapping) == list(range(n))
        edges, nodes = fig_qpu.data[0], fig_qpu.data[1]
        assert len(nodes.x) == n and len(fig_not_qpu.data) == 1 and len(fig_not_qpu.data[0].x) == n
        xs = np.asarray([v for v in edges.x if v is not None]).reshape(-1, 2)
        ys = np.asarray([v for v in edges.y if v is not None]).reshape(-1, 2)
        assert len(xs) > n and float(np.hypot(xs[:, 0] - xs[:, 1], ys[:, 0] - ys[:, 1]).max()) < 0.1
        assert len({(round(a, 6), round(b, 6)) for a, b in zip(nodes.x, nodes.y)}) == n
