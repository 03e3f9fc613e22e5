"""GPU parity of the block-Gibbs sampler and the GRBM energy kernels against the oracle
(bit-exact spins; energies within 1e-6 relative)."""
import numpy as np
import pytest
import torch

from image_generation_amd import _lib, graphs, sampler as smp
from oracle import cref, gibbs

pytestmark = pytest.mark.gpu
SEED = 775321899904


def _plan(fam, n, seed=SEED):
    g = graphs.pegasus_graph(16) if fam == "pegasus" else graphs.zephyr_graph(12)
    mg, _ = graphs.get_graph_mapping(graphs.greedy_get_subgraph(n, seed, g))
    nodes, ei, ej = graphs.edges_of(mg)
    return graphs.build_plan(n, ei, ej), nodes


def _model(plan, rng):
    h = (0.05 * rng.uniform(-1, 1, plan.n)).astype(np.float32)
    J = (5.0 * rng.uniform(-1, 1, plan.n_edges)).astype(np.float32)
    return h, J


@pytest.mark.parametrize("fam,n,C,sweeps", [("pegasus", 64, 37, 3), ("zephyr", 128, 256, 5), ("pegasus", 256, 130, 4),
                                             ("zephyr", 512, 64, 3), ("zephyr", 1024, 9, 2), ("pegasus", 128, 256, 50),
                                             ("zephyr", 512, 33, 13), ("zephyr", 1024, 41, 5)])  # last: 16-wave workgroups, ragged last one
def test_gibbs_bit_exact(fam, n, C, sweeps):
    _bit_exact(fam, n, C, sweeps)


@pytest.mark.parametrize("fam,n,C,sweeps", [("zephyr", 1024, 530, 5), ("pegasus", 1024, 513, 7), ("zephyr", 1024, 2048, 4)])
@pytest.mark.parametrize("generic", [0, 1])
def test_gibbs_bit_exact_large_graph_many_chains(fam, n, C, sweeps, generic):
    """More than 512 chains on a 1024-spin graph: the rolled schedule in 16-wave workgroups -- by default with static
    (class, pass) slots and the Philox words of a counter kept for its four sweeps (gibbs_slot_kernel; draws of 5 and 7
    sweeps start off the multiples of four), with gibbs_generic = 1 the plain rolled kernel -- against the C oracle."""
    with _lib.option_scope(gibbs_generic=generic):
        _bit_exact(fam, n, C, sweeps)


@pytest.mark.parametrize("fam,n,C,sweeps", [("zephyr", 512, 70, 6), ("pegasus", 512, 256, 5), ("zephyr", 128, 37, 9), ("zephyr", 1024, 100, 3)])
def test_gibbs_bit_exact_chains_side_by_side_in_8_chain_workgroups(fam, n, C, sweeps):
    """gibbs_generic = 4 (an A/B form: the chains-side-by-side schedule of the large graphs, gibbs_slot_kernel, in 8-chain
    workgroups on any graph whose classes make at most 20 slots) -- same bits as the oracle; ragged last workgroups."""
    with _lib.option_scope(gibbs_generic=4):
        _bit_exact(fam, n, C, sweeps)


@pytest.mark.parametrize("fam,n,C,sweeps", [("zephyr", 512, 70, 6), ("pegasus", 512, 256, 5)])
def test_gibbs_bit_exact_one_row_at_a_time(fam, n, C, sweeps):
    """Graphs whose colour classes take two passes of 64 lanes run the passes side by side by default; option
    gibbs_generic = 2 keeps the same lane-major schedule one row at a time -- same bits."""
    with _lib.option_scope(gibbs_generic=2):
        _bit_exact(fam, n, C, sweeps)


def _bit_exact(fam, n, C, sweeps):
    plan, nodes = _plan(fam, n)
    rng = np.random.default_rng(n)
    h, J = _model(plan, rng)
    s = smp.GibbsSampler(plan, nodes, beta=20.0, sweeps=sweeps, seed=SEED, persistent=True, chain_offset=1000,
                         h_range=(-4, 4), j_range=(-1, 1))
    lin = torch.from_numpy(h).cuda(); quad = torch.from_numpy(J).cuda()
    hs, Js = gibbs.scaled_fields(h, J, 0.05, (-4, 4), (-1, 1))
    ids = np.arange(C, dtype=np.uint32) + 1000
    want = cref.init_state(ids, n, SEED)
    for call in range(3):  # persistent chains over three draws
        got = s.sample_native(lin, quad, 0.05, (-4, 4), (-1, 1), num_reads=C)
        want = cref.gibbs_sweeps(want, ids, hs, Js, 20.0, plan.order, plan.class_ptr, plan.adj_ptr, plan.adj_idx,
                                 plan.adj_eid, SEED, call * sweeps, sweeps)
        g = got.cpu().numpy()
        assert g.dtype == np.float32 and set(np.unique(g).tolist()) <= {-1.0, 1.0}
        mism = int((g != want.astype(np.float32)).sum())
        assert mism == 0, f"{mism} spin mismatches at call {call}"


@pytest.mark.parametrize("n,p_edge,C,sweeps", [(77, 0.35, 19, 6), (130, 0.22, 70, 4), (61, 0.03, 5, 3)])
def test_gibbs_bit_exact_on_arbitrary_graphs(n, p_edge, C, sweeps):
    """Graphs the shipped solvers do not produce: an odd spin count (the LDS image's alignment padding), degrees beyond
    the 20 neighbours of one round of the neighbour sum (77 spins at p = 0.35: up to ~35 -> two rounds of 5 batches, last
    batches partly padding), isolated spins (rows of zero batches: the all-zero batch only), dozens of colour classes
    (the rolled schedule), and a sparse graph whose rows are single partly-padded batches -- both schedules against the C
    oracle, bit for bit."""
    rng = np.random.default_rng(n)
    iu, ju = np.triu_indices(n, 1)
    keep = rng.uniform(size=iu.size) < p_edge
    keep &= (iu != 3) & (ju != 3)                      # spin 3 is isolated
    ei, ej = iu[keep].astype(np.int64), ju[keep].astype(np.int64)
    plan = graphs.build_plan(n, ei, ej)
    deg = np.diff(plan.adj_ptr)
    assert deg[3] == 0 and (p_edge < 0.1 or deg.max() > 20)
    h = (0.5 * rng.uniform(-1, 1, n)).astype(np.float32)
    J = (2.0 * rng.uniform(-1, 1, plan.n_edges)).astype(np.float32)
    hs, Js = gibbs.scaled_fields(h, J, 0.3, (-4, 4), (-1, 1))
    ids = np.arange(C, dtype=np.uint32) + 7
    lin = torch.from_numpy(h).cuda(); quad = torch.from_numpy(J).cuda()
    for generic in (0, 1, 2):
        with _lib.option_scope(gibbs_generic=generic):
            s = smp.GibbsSampler(plan, list(range(n)), beta=1.5, sweeps=sweeps, seed=SEED, persistent=True, chain_offset=7,
                                 h_range=(-4, 4), j_range=(-1, 1))
            want = cref.init_state(ids, n, SEED)
            for call in range(2):
                got = s.sample_native(lin, quad, 0.3, (-4, 4), (-1, 1), num_reads=C).cpu().numpy()
                want = cref.gibbs_sweeps(want, ids, hs, Js, 1.5, plan.order, plan.class_ptr, plan.adj_ptr, plan.adj_idx,
                                         plan.adj_eid, SEED, call * sweeps, sweeps)
                assert int((got != want.astype(np.float32)).sum()) == 0, (generic, call)


@pytest.mark.parametrize("generic", [0, 1])
def test_gibbs_bit_exact_large_arbitrary_graph(generic):
    """A 700-spin random graph (average degree 14, some spins beyond 20 neighbours: no lane-major image, rows of up to seven
    batches, a spin count that is no multiple of 16, a dozen colour classes of uneven size): large enough for the 16-wave
    workgroups of the rolled schedule -- by default its chains-side-by-side form with the general (not five-batch) image,
    where the classes make at most 20 slots -- against the C oracle; gibbs_generic = 1: the plain rolled kernel."""
    n, C, sweeps = 700, 45, 3
    rng = np.random.default_rng(n)
    iu, ju = np.triu_indices(n, 1)
    keep = rng.uniform(size=iu.size) < 0.02
    ei, ej = iu[keep].astype(np.int64), ju[keep].astype(np.int64)
    plan = graphs.build_plan(n, ei, ej)
    assert np.diff(plan.adj_ptr).max() > 20
    h = (0.5 * rng.uniform(-1, 1, n)).astype(np.float32)
    J = (2.0 * rng.uniform(-1, 1, plan.n_edges)).astype(np.float32)
    hs, Js = gibbs.scaled_fields(h, J, 0.3, (-4, 4), (-1, 1))
    ids = np.arange(C, dtype=np.uint32) + 7
    lin = torch.from_numpy(h).cuda(); quad = torch.from_numpy(J).cuda()
    with _lib.option_scope(gibbs_generic=generic):
        s = smp.GibbsSampler(plan, list(range(n)), beta=1.5, sweeps=sweeps, seed=SEED, persistent=True, chain_offset=7,
                             h_range=(-4, 4), j_range=(-1, 1))
        want = cref.init_state(ids, n, SEED)
        for call in range(3):
            got = s.sample_native(lin, quad, 0.3, (-4, 4), (-1, 1), num_reads=C).cpu().numpy()
            want = cref.gibbs_sweeps(want, ids, hs, Js, 1.5, plan.order, plan.class_ptr, plan.adj_ptr, plan.adj_idx,
                                     plan.adj_eid, SEED, call * sweeps, sweeps)
            assert int((got != want.astype(np.float32)).sum()) == 0, (generic, call)


def test_gibbs_bit_exact_many_small_classes():
    """50 disjoint 14-cliques: 14 colour classes of 50 spins -> one pass of 64 lanes each, 14 rows per lane -- the
    lane-major schedule's one-row form at more than 12 rows (no shipped graph has that shape), against the C oracle."""
    k, m = 14, 50
    n = k * m
    ei, ej = [], []
    for c in range(m):
        for a in range(k):
            for b in range(a + 1, k):
                ei.append(c * k + a); ej.append(c * k + b)
    plan = graphs.build_plan(n, np.asarray(ei, np.int64), np.asarray(ej, np.int64))
    sizes = np.diff(plan.class_ptr)
    assert plan.n_colours == k and sizes.max() == m
    rng = np.random.default_rng(7)
    h = (0.5 * rng.uniform(-1, 1, n)).astype(np.float32)
    J = (1.0 * rng.uniform(-1, 1, plan.n_edges)).astype(np.float32)
    hs, Js = gibbs.scaled_fields(h, J, 0.4, (-4, 4), (-1, 1))
    C, sweeps = 21, 5
    ids = np.arange(C, dtype=np.uint32) + 3
    lin = torch.from_numpy(h).cuda(); quad = torch.from_numpy(J).cuda()
    for generic in (0, 1):
        with _lib.option_scope(gibbs_generic=generic):
            s = smp.GibbsSampler(plan, list(range(n)), beta=1.2, sweeps=sweeps, seed=SEED, persistent=True, chain_offset=3,
                                 h_range=(-4, 4), j_range=(-1, 1))
            want = cref.init_state(ids, n, SEED)
            for call in range(2):
                got = s.sample_native(lin, quad, 0.4, (-4, 4), (-1, 1), num_reads=C).cpu().numpy()
                want = cref.gibbs_sweeps(want, ids, hs, Js, 1.2, plan.order, plan.class_ptr, plan.adj_ptr, plan.adj_idx,
                                         plan.adj_eid, SEED, call * sweeps, sweeps)
                assert int((got != want.astype(np.float32)).sum()) == 0, (generic, call)


def test_sample_ising_dict_path_and_sampleset():
    plan, nodes = _plan("pegasus", 64)
    rng = np.random.default_rng(1)
    h, J = _model(plan, rng)
    hs, Js = gibbs.scaled_fields(h, J, 0.05, (-4, 4), (-1, 1))
    s = smp.GibbsSampler(plan, nodes, beta=20.0, sweeps=2, seed=5, persistent=False)
    hd = {v: float(hs[k]) for k, v in enumerate(nodes)}
    Jd = {(nodes[a], nodes[b]): float(Js[e]) for e, (a, b) in enumerate(zip(plan.edge_i, plan.edge_j))}
    ss = s.sample_ising(hd, Jd, num_reads=16, answer_mode="raw", auto_scale=False, annealing_time=1, label="x")
    ids = np.arange(16, dtype=np.uint32)
    want = cref.gibbs_sweeps(cref.init_state(ids, 64, 5), ids, hs, Js, 20.0, plan.order, plan.class_ptr,
                             plan.adj_ptr, plan.adj_idx, plan.adj_eid, 5, 0, 2)
    assert np.array_equal(ss.record.sample, want)
    assert ss.variables == nodes and ss.vartype == "SPIN" and len(ss) == 16
    # non-persistent: the second draw restarts the chains from a configuration keyed by its first sweep index (2), not
    # from the first draw's start
    ss2 = s.sample_ising(hd, Jd, num_reads=16, answer_mode="raw", auto_scale=False, annealing_time=1, label="x")
    start2 = cref.init_state(ids, 64, 5, sweep0=2)
    assert not np.array_equal(start2, cref.init_state(ids, 64, 5))
    want2 = cref.gibbs_sweeps(start2, ids, hs, Js, 20.0, plan.order, plan.class_ptr, plan.adj_ptr, plan.adj_idx,
                              plan.adj_eid, 5, 2, 2)
    assert np.array_equal(ss2.record.sample, want2)


def test_shim_composite_draw_matches_oracle():
    """dwave.system shim: DWaveSampler(name) + FixedEmbeddingComposite(one-to-one embedding).sample_ising draws with the
    GPU sampler on the induced sub-graph -- bit-exact against the C restatement."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "image-generation_amd", "shims"))
    try:
        import dwave.system as dsys
        from image_generation_amd import graphs
        dsys.LOCAL_SOLVER.update(sweeps=3, beta=20.0, seed=77, persistent=False)
        qpu = dsys.DWaveSampler(solver="Advantage_system4")
        sub = graphs.greedy_get_subgraph(64, 11, qpu.to_networkx_graph())
        _mapped, mapping = graphs.get_graph_mapping(sub)  # physical -> logical 0..63
        comp = dsys.FixedEmbeddingComposite(qpu, {l_: [p] for p, l_ in mapping.items()})
        plan = comp._build().plan
        rng = np.random.default_rng(5)
        h, J = _model(plan, rng)
        hs, Js = gibbs.scaled_fields(h, J, 0.05, (-4, 4), (-1, 1))
        hd = {k: float(hs[k]) for k in range(64)}
        Jd = {(int(a), int(b)): float(Js[e]) for e, (a, b) in enumerate(zip(plan.edge_i, plan.edge_j))}
        ss = comp.sample_ising(hd, Jd, num_reads=16, answer_mode="raw", auto_scale=False, annealing_time=1, label="x")
        ids = np.arange(16, dtype=np.uint32)
        want = cref.gibbs_sweeps(cref.init_state(ids, 64, 77), ids, hs, Js, 20.0, plan.order, plan.class_ptr, plan.adj_ptr,
                                 plan.adj_idx, plan.adj_eid, 77, 0, 3)
        assert ss.variables == list(range(64)) and ss.vartype == "SPIN"
        assert np.array_equal(ss.record.sample.astype(np.int8), want.astype(np.int8))
    finally:
        sys.path.pop(0)
        for name in [m for m in sys.modules if m.split(".")[0] in ("dwave", "dimod")]:
            del sys.modules[name]


def test_energy_and_suffstats():
    plan, nodes = _plan("zephyr", 128)
    rng = np.random.default_rng(3)
    h, J = _model(plan, rng)
    x = np.where(rng.random((300, 128)) < 0.5, -1.0, 1.0).astype(np.float32)
    x[:7] = rng.standard_normal((7, 128)).astype(np.float32)  # general floats too
    L = _lib.lib()
    gh = smp.GraphHandle(plan, "cuda")
    xd = torch.from_numpy(x).cuda(); hd = torch.from_numpy(h).cuda(); Jd = torch.from_numpy(J).cuda()
    e = torch.empty(300, device="cuda")
    st = _lib.stream_ptr()
    _lib.check(L.dvg_grbm_energy(gh.ptr, xd.data_ptr(), 300, hd.data_ptr(), Jd.data_ptr(), e.data_ptr(), st))
    want = x.astype(np.float64) @ h + (x[:, plan.edge_i] * x[:, plan.edge_j]).astype(np.float64) @ J
    np.testing.assert_allclose(e.cpu().numpy(), want, rtol=1e-6, atol=1e-5)
    ws = torch.empty(L.dvg_grbm_suffstats_workspace_bytes(gh.ptr), dtype=torch.uint8, device="cuda")
    gl = torch.zeros(128, device="cuda"); gq = torch.zeros(plan.n_edges, device="cuda")
    w = torch.from_numpy(rng.standard_normal(300).astype(np.float32)).cuda()
    _lib.check(L.dvg_grbm_suffstats(gh.ptr, xd.data_ptr(), 300, None, 1.0 / 300, gl.data_ptr(), gq.data_ptr(), 0,
                                    ws.data_ptr(), ws.numel(), st))
    np.testing.assert_allclose(gl.cpu().numpy(), x.mean(0), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(gq.cpu().numpy(), (x[:, plan.edge_i] * x[:, plan.edge_j]).mean(0), rtol=1e-6, atol=1e-6)
    _lib.check(L.dvg_grbm_suffstats(gh.ptr, xd.data_ptr(), 300, w.data_ptr(), -2.0, gl.data_ptr(), gq.data_ptr(), 1,
                                    ws.data_ptr(), ws.numel(), st))
    wn = w.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(gl.cpu().numpy(), x.mean(0) - 2 * (wn[:, None] * x).sum(0), rtol=1e-5, atol=1e-5)


def test_invalid_colouring_rejected():
    plan, nodes = _plan("pegasus", 64)
    bad = graphs.GibbsPlan(**{**plan.__dict__})
    bad.class_ptr = np.array([0, 64], dtype=np.int32)  # one class containing edges
    with pytest.raises(_lib.DvgError, match="colouring"):
        smp.GraphHandle(bad, "cuda")
