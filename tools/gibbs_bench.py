"""The block-Gibbs draw ALONE (nothing else on the chip): `python tools/gibbs_bench.py [n chains sweeps]` on an MI355X.
Prints the draw time per kernel form -- inside a training step the same draw shares its CUs
with the encoder's GEMMs and takes longer (bench.py's `sampler` entry is that in-situ number)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from image_generation_amd import _lib, graphs, sampler as smp

n, C, sweeps = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 256, 200)))
seed = 775321899904
topology = graphs.pegasus_graph(16) if "pegasus" in sys.argv else graphs.zephyr_graph(12)  # (bench.py: c1 / c2 Pegasus, c3 / c5 Zephyr)
mg, _ = graphs.get_graph_mapping(graphs.greedy_get_subgraph(n, seed, topology))
nodes, ei, ej = graphs.edges_of(mg)
plan = graphs.build_plan(n, ei, ej)
rng = np.random.default_rng(1)
lin = torch.from_numpy((0.05 * rng.uniform(-1, 1, n)).astype(np.float32)).cuda()
quad = torch.from_numpy((5.0 * rng.uniform(-1, 1, plan.n_edges)).astype(np.float32)).cuda()
print(f"n={n} chains={C} sweeps={sweeps} colours={plan.n_colours} max class={max(np.diff(plan.class_ptr))} "
      f"max degree={int(np.diff(plan.adj_ptr).max())}")
only_default = "default" in sys.argv  # (PMC passes: the product's form only)
forms = (("default", dict()), ("one row at a time", dict(gibbs_generic=2)), ("rolled reference", dict(gibbs_generic=1)),
         ("chains side by side, 8-chain workgroups", dict(gibbs_generic=4)))
for name, opts in forms[:1] if only_default else forms:
    with _lib.option_scope(**opts):
        s = smp.GibbsSampler(plan, nodes, beta=20.0, sweeps=sweeps, seed=seed, persistent=True)
        s.sample_native(lin, quad, 0.05, (-4, 4), (-1, 1), num_reads=C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = s.sample_native(lin, quad, 0.05, (-4, 4), (-1, 1), num_reads=C)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5
        print(f"  {name:26s} {t*1e3:9.1f} us per draw  {C*n*sweeps/t/1e6:8.2f} G spin updates/s  checksum {float(out.sum()):.0f}")
