#!/usr/bin/env python3
"""Headline benchmark: DVAE + GRBM train-step images/s on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full `ModelWrapper.step` (/root/reference/src/model_wrapper.py:279-353) on one
synthetic batch resident in HBM: encoder fwd+bwd, Gumbel discretisation, R decoder replicas fwd+bwd,
MSE, one block-Gibbs draw, fused MMD fwd+bwd, Adam; the GRBM quasi-NLL branch (second draw, energy,
sufficient statistics, Adam) runs at its natural duty of every 10th step.  Prints ONE JSON line.

Workload: BASELINE.json's metric names no configuration, so the N=1 line is the LARGEST single-GPU configuration,
c3 (configs[2]: B=4096, 512-spin Zephyr GRBM, 200-sweep PCD); c2 and c1 are timed by child runs and reported under
`extra`.  At N>1 every rank runs the c3 workload (= configs[3] at N=8), weak scaling.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import yaml  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[0]: the reference's own CPU-runnable case
    "c1": dict(B=64, n=64, R=8, C=256, sweeps=1, qpu="Advantage_system4", persistent=False,
               desc="MNIST-shaped 32x32 synthetic, B=64, 64-spin Pegasus sub-graph GRBM, R=8, 256 reads, 1 Gibbs sweep"),
    # configs[1]
    "c2": dict(B=256, n=128, R=8, C=256, sweeps=50, qpu="Advantage_system4", persistent=True,
               desc="MNIST-shaped 32x32 synthetic, B=256, 128-spin Pegasus sub-graph GRBM, R=8, 256 reads, 50-sweep PCD Gibbs"),
    # configs[2]: the largest single-GPU configuration -> the bench workload
    "c3": dict(B=4096, n=512, R=8, C=256, sweeps=200, qpu="Advantage2_system1", persistent=True,
               desc="MNIST-shaped 32x32 synthetic, B=4096, 512-spin Zephyr sub-graph GRBM, R=8, 256 reads, 200-sweep PCD Gibbs"),
    # configs[3] is c3 per GPU on 8 GPUs: `--config c3 --gpus 8`.  configs[4], per-GPU slice (16384 chains / 8 GPUs):
    # the sampler-bound case (every step draws 2048 chains of 1024 spins; the MMD sees 2048 x 2048 rows of d = 1024)
    "c5": dict(B=256, n=1024, R=8, C=2048, sweeps=50, qpu="Advantage2_system1", persistent=True,
               desc="Fashion-MNIST-shaped 32x32 synthetic, B=256, 1024-spin Zephyr sub-graph GRBM, R=8, 2048 chains per GPU, "
                    "50-sweep PCD Gibbs"),
}
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense (the spin-path MMD gradient GEMM runs there)
PEAK_HBM_GBS = 8000.0
N_CUS, CLOCK_GHZ = 256, 2.4  # MI355X_MICROARCH.md chip-level parameters


def baseline_metric():
    """BASELINE.json's own wording of the metric (the file ships with the repository); a fixed id otherwise."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "dvae_grbm_train_step_images_per_s"


def write_yaml(cfg, path, precision="f32"):
    base = yaml.safe_load(open(os.path.join(ROOT, "image-generation_amd", "training_parameters.yaml")))
    base.update(BATCH_SIZE=cfg["B"], N_REPLICAS=cfg["R"], NUM_READS=cfg["C"], GIBBS_SWEEPS=cfg["sweeps"],
                GIBBS_PERSISTENT=cfg["persistent"], CONV_PRECISION=precision)
    with open(path, "w") as f:
        yaml.safe_dump(base, f)


def net_flops_per_image(n, R):
    f_enc = 18 * (32 * 1024 + 32 * 64 * 256 + 64 * 128 * 64 + 128 * n * 16) + 8 * n
    f_dec = 8 * n * n + 18 * (n * 128 * 4 + 128 * 64 * 16 + 64 * 32 * 64 + 32 * 256 + 1024)
    return 3 * (f_enc + R * f_dec)


# (rocprof spells out the kernels' trailing arithmetic-form template argument; the float32 instantiations -- 3, the LDS-DMA
# form the library launches, or 0, the register-staged one -- are the ones priced)
_NORM = lambda s_: s_.replace(" ", "").replace("voiddvg::", "").replace("dvg::", "").split("(")[0].replace(",false>", ">").replace(",1,0>", ",1>").replace(",1,3>", ",1>")  # noqa: E731
# name of the rocprof kernel behind a library profiler id that is not itself a kernel name
PROF_TO_ROCPROF = {"mmd_pm1": "mmd_pair", "gibbs_sweeps": "gibbs_", "conv_wino_kernel": "conv_wino8_kernel",
                   "conv_wino_wgrad_kernel": "conv_wino_wgrad8_kernel"}


def _profile(kind, config, lib_hash):
    """A committed profiles/ JSON of this round, or None when it was measured on a different build of the kernels:
    every file carries `kernels_hash` = dvg_source_hash() of the library it was measured on (tools/make_profiles.sh),
    and a number from another build would survive a kernel regression unchanged."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{kind}_{config}.json")
        if os.path.exists(path):
            d = json.load(open(path))
            if d.get("kernels_hash") == lib_hash:
                return d, os.path.relpath(path, ROOT)
            return None, f"{os.path.relpath(path, ROOT)}: stale (kernels_hash {d.get('kernels_hash')} != library {lib_hash})"
    return None, None


def _find_kernel(d, kernel):
    want = PROF_TO_ROCPROF.get(kernel, kernel)
    best = None
    for name, rec in d.get("kernels", {}).items():
        if _NORM(name) == _NORM(want) or _NORM(name).startswith(_NORM(want)):
            weight = lambda r: r.get("launches", 0) * r.get("avg_us_alone", r.get("dispatch_cycles_avg", 1.0))  # noqa: E731
            if best is None or weight(rec) > weight(best):
                best = rec
    return best


def pmc_traffic(kernel, config, lib_hash):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE and
    --pmc WRITE_SIZE runs of `bench.py --eager`; KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).
    PMC counters cannot be read from inside the process, so the number is the last committed measurement -- of THIS
    build of the kernels (hash-checked), else null."""
    d, src = _profile("pmc_traffic", config, lib_hash)
    rec = _find_kernel(d, kernel) if d else None
    return (rec["hbm_bytes_per_launch"] if rec else None), src


def pmc_mfma_busy(kernel, config, lib_hash):
    """Matrix-pipe utilisation of `kernel` (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)) from the
    committed rocprofv3 PMC pass of `bench.py --eager` (profiles/aggregate_mfma.py): dispatches are serialised under
    PMC collection, so this is the kernel alone, launch ramp included.  Hash-checked like the traffic."""
    d, _src = _profile("pmc_mfma_busy", config, lib_hash)
    rec = _find_kernel(d, kernel) if d else None
    return rec["mfma_busy_frac"] if rec else None


def pmc_mfma_busy_all(kernel, config, lib_hash):
    """... of EVERY instantiation the profiler id `kernel` covers (conv_wino4_kernel<2>, <3>, <4>: one per image size),
    {rocprof name: (matrix-pipe busy fraction alone at the in-step grid, launches per profiled run)}."""
    d, _src = _profile("pmc_mfma_busy", config, lib_hash)
    if not d:
        return None
    want = _NORM(PROF_TO_ROCPROF.get(kernel, kernel))
    out = {name.split("(")[0].replace("void dvg::", ""): {"mfma_busy_frac": rec["mfma_busy_frac"], "launches": rec["launches"]}
           for name, rec in d.get("kernels", {}).items() if _NORM(name) == want or _NORM(name).startswith(want)}
    return out or None


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_leg(config, threads, warm, min_steps, seconds, natural_grbm):
    """(child process of cpu_baseline: `bench.py --cpu-leg CONFIG --threads T ...`)  Times the CPU oracle on `threads`
    threads of a FRESH process -- stock PyTorch's OpenMP pool does not shrink cleanly inside one process (measured on the
    GPU box: after a 64-thread point the 32-thread steps ran 3x slower than in a process that only ever had 32) -- and
    prints one JSON object."""
    from image_generation_amd import graphs
    from oracle.step import OracleTrainer

    cfg = CONFIGS[config]
    torch.set_num_threads(threads)
    make, h_range, j_range = graphs.LOCAL_SOLVERS[cfg["qpu"]]
    mg, _ = graphs.get_graph_mapping(graphs.greedy_get_subgraph(cfg["n"], 775321899904, make()))
    _, ei, ej = graphs.edges_of(mg)
    plan = graphs.build_plan(cfg["n"], ei, ej)
    tr = OracleTrainer(plan, cfg["n"], cfg["R"], cfg["C"], cfg["sweeps"], 0.05, seed=1, h_range=h_range, j_range=j_range)
    Bc = min(cfg["B"], 512)
    g = torch.Generator().manual_seed(3)
    batch = lambda: (torch.rand((Bc, 1, 32, 32), generator=g) < 0.13).float()  # noqa: E731
    for _ in range(warm):
        tr.step(batch(), force_grbm=False)
    times = []
    t_all = time.perf_counter()
    while len(times) < min_steps or (time.perf_counter() - t_all < seconds and len(times) < 40):
        t0 = time.perf_counter()
        if natural_grbm:
            tr.step(batch())  # GRBM branch at its natural duty (step 10k)
        else:
            tr.step(batch(), force_grbm=False)
        times.append(time.perf_counter() - t0)
    times.sort()
    print(json.dumps({"threads": threads, "batch": Bc, "times_ms": [t * 1e3 for t in times]}), flush=True)


def cpu_baseline(config, seconds=20.0, sweep=(8, 16, 32, 64)):
    """The CPU oracle (a port: stock PyTorch CPU ops + the C Gibbs restatement) on this box's host cores, on a BOUNDED
    sample of the workload: the same model, sampler and replica count at a batch of at most 512 images per step (a c3
    step is 4096; the oracle's MMD materialises the (B R + C)^2 kernel matrix as the reference does, 4.4 GB per
    temporary at c3), 3 warm-up steps, then >= 10 timed steps for about `seconds`; median step time.
    The sample's batch is NOT the GPU line's batch when the workload's is above 512: the oracle's MMD is quadratic in
    B R, so images/s at B = 512 flatters the CPU against the same model at B = 4096 -- `batch` and
    `comparable_to_value` say so in the line (c1, B = 64, runs at its own size: comparable).

    `cores`: stock PyTorch's intra-op pool does not scale on these layer sizes, so the thread count is MEASURED: a short
    sweep (1 warm-up + 3 timed steps per point, each point a fresh process) over `sweep`, capped at the host's core
    count; 16 threads run the baseline proper (again a fresh process) unless another point is more than 20 % faster, and
    the sweep is recorded in the line (`thread_sweep`)."""
    import subprocess

    cfg = CONFIGS[config]
    host = os.cpu_count() or 1
    points = sorted({min(t, host) for t in sweep}) or [min(host, 16)]

    def leg(threads, warm, min_steps, secs, natural):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", config, "--threads", str(threads), "--leg-warm", str(warm),
               "--leg-steps", str(min_steps), "--leg-seconds", str(secs)] + (["--leg-natural"] if natural else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"threads"')]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f"cpu leg ({threads} threads) failed with code {r.returncode}: {r.stderr[-400:]}")
        return json.loads(lines[-1])

    thread_sweep = {}
    for t in points:
        d = leg(t, 1, 3, 0.0, False)
        thread_sweep[str(t)] = {"ms_per_step": d["times_ms"][len(d["times_ms"]) // 2], "images_per_s": d["batch"] / (d["times_ms"][len(d["times_ms"]) // 2] * 1e-3)}
    # (the baseline runs on the FASTEST point of the sweep -- a preference for 16 threads within 20 % biased the
    # denominator of the reported speed-up in the GPU's favour; `cores_fastest_in_sweep` is kept beside `cores` so that a
    # reader sees when a noisy 3-step point decided)
    best = min(thread_sweep, key=lambda k: thread_sweep[k]["ms_per_step"])
    cores = int(best)
    d = leg(cores, 3, 10, seconds, True)
    times, Bc = d["times_ms"], d["batch"]
    med = times[len(times) // 2] * 1e-3
    return {"value": Bc / med, "unit": "images/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "host_cores": host, "batch": Bc, "workload_batch": cfg["B"], "timed_steps": len(times),
            "comparable_to_value": bool(Bc == cfg["B"]),
            "thread_sweep": thread_sweep,
            "sample": f"{len(times)} train steps of the {cfg['B']}-image workload's model (n={cfg['n']}, R={cfg['R']}, "
                      f"{cfg['C']} chains x {cfg['sweeps']} sweeps) at B={Bc} images per step on the CPU oracle after 3 "
                      f"warm-up steps, {cores} threads in a fresh process (the fastest of {points}, each measured in its own "
                      f"process over 3 steps); median {med * 1e3:.0f} ms/step, min {times[0]:.0f} ms",
            "ms_per_step_median": med * 1e3, "ms_per_step_min": times[0]}


def loss_parity(side_mode=None):
    """The "loss parity vs CPU" half of the metric: 12 training steps on this GPU against the committed fixture of the
    reference's own ``ModelWrapper.step`` driven over the CPU oracle on identical batches, Gumbel noise and dropout
    masks (tests/golden/make_golden.py), and a sampler draw against the C restatement.  Checker only: the timed
    region above never touches it.  `side_mode` ("f32x3" / "bf16"): that operand mode's losses at each of the fixture's
    12 training states, taken from the float32 trajectory's saved state (``parity_check_by_state``: a free run in a side
    mode compares chaotic trajectories -- profiles/r05_fixture_knife_edge.txt), with the free run's figures beside."""
    import __graft_entry__ as entry

    r = entry.parity_check(steps=12)
    if side_mode:
        s = entry.parity_check_by_state(side_mode, steps=12)
        # (the by-state losses are each step's FORWARD pass only -- named so -- and the side mode's backward
        # pass is reported beside them: every parameter gradient against the float32 step from the same state)
        return {"steps": s["steps"], "max_rel_dev_by_state_forward_only": s["max_rel_dev"], "tolerance": 1e-5, "mode": side_mode,
                "how": "each step taken from the float32 trajectory's saved training state",
                "grad_rel_l2_vs_f32_by_state": {k: s["grad_rel_l2_vs_f32_by_state"][k] for k in ("worst_tensor", "worst")},
                "max_rel_dev_free_run": r["max_rel_dev"], "gibbs_spin_mismatches": r["gibbs_spin_mismatches"],
                "gibbs_spins_checked": r["gibbs_spins_checked"],
                "against": "tests/golden/step_n64.npz (reference step orchestration over the CPU oracle, B=8, n=64, R=2)"}
    return {"steps": r["steps"], "max_rel_dev": r["max_rel_dev"], "tolerance": 1e-5,
            "gibbs_spin_mismatches": r["gibbs_spin_mismatches"], "gibbs_spins_checked": r["gibbs_spins_checked"],
            "against": "tests/golden/step_n64.npz (reference step orchestration over the CPU oracle, B=8, n=64, R=2)"}


def bf16_inputs_run(args):
    """The same workload with bf16 operands in the forward / data-gradient convolution GEMMs (f32 accumulate), timed by
    a child run of this script after the float32 line is complete: BASELINE.json's configs[1] names bf16, the parity
    bar of the north star needs float32 -- both are reported, the bench `value` is the float32 one."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--precision", "bf16", "--no-cpu-baseline", "--child"] + (["--eager"] if args.eager else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
        d = json.loads(lines[-1])
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
                "dominant_gemm": {k: d["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac")},
                "numerics": "forward / data-gradient GEMMs equal float32 convolutions of bf16-rounded operands "
                            "(tests/test_gpu_kernels.py); losses then differ from the float32 path at the 1e-2 level"}
    except Exception as exc:  # the float32 line must not depend on the extra run
        return {"error": repr(exc)}


def split3_run(args):
    """The same workload with the operands of the LARGE forward / data-gradient launches carried as three bf16 pieces each
    (DVG_PRECISION_F32_SPLIT3: six exact piece products per k-step on the bf16 MFMA, float32 accumulation -- float32-class
    results, see `loss_parity` inside), timed by a child run of this script.  Reported BESIDE the headline, which stays on
    the f32 MFMA."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--precision", "f32x3", "--no-cpu-baseline", "--child", "--parity"] + (["--eager"] if args.eager else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
        d = json.loads(lines[-1])
        rf = d["roofline"]
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
                "dominant_kernel": {k: rf.get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_us", "pricing")},
                "conv_all": rf.get("conv_all"), "loss_parity": d.get("loss_parity")}
    except Exception as exc:
        return {"error": repr(exc)}


def extra_config_run(args, config):
    """The other single-GPU configurations (c2 = configs[1], c1 = configs[0]), timed by a child run of this script after
    the headline line is complete; reported under `extra`, never as `value`."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--config", config, "--steps", "30", "--warmup", "5", "--child",
           "--no-cpu-baseline"]

    def child(extra):
        r = subprocess.run(cmd + extra, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f"child run {extra} failed with code {r.returncode}: {r.stderr[-300:]}")
        return json.loads(lines[-1])

    try:
        d = child([])
        rf = d["roofline"]
        out = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
               "dtype": d["dtype"],
               "dominant_kernel": {k: rf.get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_us")},
               "sampler": rf.get("sampler")}
    except Exception as exc:  # the headline line must not depend on the extra runs
        return {"error": repr(exc)}
    if config == "c2":
        # BASELINE.json configs[1] names c2 "bf16"; SURVEY 8d: "bf16 GEMM inputs / f32 accumulate ... report both" -- the
        # same workload in the two other operand modes, beside the float32 number above (never the bench `value`)
        for key, prec in (("bf16_inputs", "bf16"), ("f32x3", "f32x3")):
            try:
                dd = child(["--precision", prec])
                out[key] = {"value": dd["value"], "unit": dd["unit"], "ms_per_step": dd["ms_per_step"], "dtype": dd["dtype"],
                            "dominant_kernel": {k: dd["roofline"].get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_us")}}
            except Exception as exc:
                out[key] = {"error": repr(exc)}
    return out


def layerwise_run(args):
    """The same workload with the decoder's first two layers run one by one as the reference writes them -- Linear(n, 4n),
    then the 9-tap ConvTranspose GEMM over the 2x2 images -- instead of the library's default (ONE composed linear map
    per image, padding taps never multiplied: DESIGN.md 3).  Same function, same gradients of every parameter; reported
    beside the headline so that the gain of the reformulation stays visible."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--no-cpu-baseline", "--child"]
    try:
        r = subprocess.run(cmd + ["--option", "dec_lc0=0", "--option", "dec_d22=0"], capture_output=True, text=True, timeout=900)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
        d = json.loads(lines[-1])
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                "net_gflop_per_step": d["config"].get("net_gflop_per_step"),
                "note": "options dec_lc0 = 0, dec_d22 = 0: Linear and the first ConvTranspose layer as two layer-by-layer GEMMs (9 taps)"}
    except Exception as exc:  # the headline line must not depend on the extra runs
        return {"error": repr(exc)}


def sampler_roofline(per_kernel, cfg, plan, prof_steps, config, lib_hash):
    """SURVEY.md 8d for the block-Gibbs draw: spin updates/s, the bound that actually binds it, and the algorithmic HBM
    bytes per draw against its duration (the chain state lives in LDS for all sweeps of a draw, so HBM traffic is the
    tables in and the state / samples out, once per draw).

    The bound is instruction ISSUE (rounds 1-3 printed an LDS-operation model here and DESIGN.md admitted that it was not
    what binds the kernel): one wave per SIMD issues in program order, at best one vector / LDS / scalar instruction per
    4 cycles (MI355X_MICROARCH.md, "vector-instruction ISSUE cost": one wave's stream on one SIMD), two cycles of the
    SIMD-32 per wave64 VALU instruction where waves share a SIMD.  The instruction counts are MEASURED: rocprofv3 PMC pass of
    the draw alone (SQ_INSTS_VALU / SALU / LDS / SMEM, SQ_WAVES; tools/make_profiles.sh -> profiles/rNN_pmc_gibbs_insts_*.json,
    hash-stamped like the other PMC files: a count from another build of the kernel is not used).  floor = max(wave-
    instructions per wave x 4, VALU wave-instructions per SIMD x 2) cycles at 2.4 GHz; `issue_bound_frac` = floor / the
    in-situ draw time.  What is left of 1 is latency the kernel does not hide (colour classes are dependent phases); the
    floor itself only moves by issuing fewer instructions per spin update."""
    g = per_kernel.get("gibbs_sweeps")
    if not g or not g["work"]:
        return None
    deg = 2.0 * plan.n_edges / plan.n
    rate = g["work"] / (g["total_ms"] * 1e-3)
    n, ne, C = plan.n, plan.n_edges, cfg["C"]
    hbm = (n + ne) * 4 + 2 * ne * 4 + 2 * ne * 2 + C * n * 2 + C * n * 4  # h, J, edge ids, CSR indices; state r+w; f32 samples
    per_draw_s = g["total_ms"] * 1e-3 / g["launches"]
    out = {"kernel": "gibbs_fast_kernel / gibbs_kernel", "spin_updates_per_s": rate, "avg_draw_us": per_draw_s * 1e6,
           "draws_per_step": g["launches"] / prof_steps, "spin_updates_per_draw": g["work"] / g["launches"],
           "bound": "instruction issue", "mean_degree": deg,
           "hbm_bytes": hbm, "hbm_GBps": hbm / per_draw_s / 1e9, "hbm_frac_of_peak": hbm / per_draw_s / 1e9 / PEAK_HBM_GBS}
    d, src = _profile("pmc_gibbs_insts", config, lib_hash)
    out["issue_bound_source"] = src
    rec = None
    if d and (d.get("n"), d.get("chains"), d.get("sweeps")) == (n, C, cfg["sweeps"]):
        recs = [r for k, r in d.get("kernels", {}).items() if "gibbs" in k]
        rec = max(recs, key=lambda r: r["launches"]) if recs else None
    if rec:
        ins = rec["insts_per_launch"]
        total = sum(ins.get(k, 0.0) for k in ("valu", "salu", "lds", "smem"))
        per_wave = total / rec["waves"]
        simds = min(1024.0, 4.0 * rec["cus_used"])
        floor_cycles = max(per_wave * 4.0, ins.get("valu", 0.0) / simds * 2.0)
        floor_s = floor_cycles / (CLOCK_GHZ * 1e9)
        out.update({"issue_bound_draw_us": floor_s * 1e6, "issue_bound_frac": floor_s / per_draw_s,
                    "issue_bound_spin_updates_per_s": g["work"] / g["launches"] / floor_s,
                    "insts_per_wave_per_sweep": rec["insts_per_wave_per_sweep"],
                    "insts_per_spin_update": {"valu": rec["valu_per_spin_update"], "lds": rec["lds_per_spin_update"]},
                    "waves": rec["waves"], "waves_per_simd": rec["waves_per_simd"], "cus_used": rec["cus_used"],
                    "draw_us_alone_under_pmc": rec["avg_us_alone_under_pmc"]})
    else:
        out.update({"issue_bound_draw_us": None, "issue_bound_frac": None})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--child", action="store_true", help="(internal) a child run of another configuration: no extras")
    ap.add_argument("--parity", action="store_true", help="(internal) run the 12-step loss-parity check in this arithmetic mode")
    ap.add_argument("--eager", action="store_true", help="no hipGraph replay: every step launches its kernels one by one")
    ap.add_argument("--breakdown", default="", help="write the per-kernel HIP-event breakdown (JSON) to this path")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="a kernel-form option of the library (dvg_set_option), e.g. dec_lc0=0; may be repeated")
    ap.add_argument("--batch", type=int, default=0, help="(with --child) override the configuration's batch size")
    ap.add_argument("--wrapper", action="append", default=[], metavar="ATTR=VALUE",
                    help="set a ModelWrapper attribute before the run (A/B measurements; not for the reported line)")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16", "f32x3"],
                    help="operands of the forward / data-gradient convolution GEMMs: f32 (the 1e-5 loss parity; the bench "
                         "line) or bf16 inputs with f32 accumulate (BASELINE.json configs[1] names bf16; reported beside it)")
    ap.add_argument("--cpu-leg", default="", help="(internal) time the CPU oracle on CONFIG in this process and print JSON")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--leg-warm", type=int, default=3)
    ap.add_argument("--leg-steps", type=int, default=10)
    ap.add_argument("--leg-seconds", type=float, default=20.0)
    ap.add_argument("--leg-natural", action="store_true")
    args = ap.parse_args()
    if args.cpu_leg:
        _cpu_leg(args.cpu_leg, args.threads, args.leg_warm, args.leg_steps, args.leg_seconds, args.leg_natural)
        return
    cfg = CONFIGS[args.config]
    if args.batch > 0:  # size studies only (tools/): the reported line is always the named configuration
        if not args.child:
            raise SystemExit("--batch is for --child measurement runs")
        cfg = CONFIGS[args.config] = dict(cfg, B=args.batch, desc=cfg["desc"] + f" [B overridden: {args.batch}]")

    from image_generation_amd import _lib
    from image_generation_amd.data import synthetic_images
    from image_generation_amd.model_wrapper import ModelWrapper
    from image_generation_amd.parallel import DataParallel

    dp = DataParallel()
    if dp.world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={dp.world_size}: launch with torch.distributed.run")
    if dp.active:
        import torch.distributed as tdist

        if tdist.get_world_size() != args.gpus and not (dp.force and args.gpus == 1):
            raise SystemExit(f"--gpus {args.gpus} but the process group has {tdist.get_world_size()} ranks")
    dev = torch.device("cuda", dp.local_rank)
    torch.cuda.set_device(dev)

    tmp = tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False)
    tmp.close()
    write_yaml(cfg, tmp.name, args.precision)
    model = ModelWrapper(cfg["qpu"], n_latents=cfg["n"], training_parameter_file=tmp.name,
                         dist=dp if (dp.world_size > 1 or dp.force) else None)
    # synthetic batches resident in HBM (a pool of 64, SURVEY.md 8d: no step re-reads a batch that is still in a cache --
    # c3: 64 x 16.8 MB = 1.07 GB, four times the 256 MiB Infinity Cache)
    pool = 64
    imgs = synthetic_images(pool * cfg["B"], seed=775321899904 + dp.rank, device=dev).reshape(pool, cfg["B"], 1, 32, 32)
    labels = torch.zeros(cfg["B"], dtype=torch.int64, device=dev)
    batches = [(imgs[k], labels) for k in range(pool)]
    model.set_dataloader(batches * ((args.steps + args.warmup + 64) // pool + 2))
    model.train_init(n_epochs=1)
    model.sync_losses = False  # no .item() host syncs inside the step
    model.keep_step_losses = False  # ... and no per-step copies of the loss scalars into the history lists

    L = _lib.lib()
    for item in args.option:
        name, value = item.split("=")
        _lib.set_option(name, int(value, 0))
    _lib.check(L.dvg_set_conv_precision({"f32": 0, "bf16": 1, "f32x3": 2}[args.precision]), "dvg_set_conv_precision")
    names = [L.dvg_prof_kernel_name(i).decode() for i in range(L.dvg_prof_num_kernels())]
    is_gemm = lambda nm: nm.startswith("conv_igemm") or nm.startswith("conv_wgrad_kernel") or nm in ("mmd_main", "mmd_pm1", "conv_wgrad_fold_kernel", "conv_wino_kernel", "conv_wino_wgrad_kernel", "conv_wino4_kernel", "conv_wino4_wgrad_kernel")  # noqa: E731
    mask = sum(1 << i for i, nm in enumerate(names) if is_gemm(nm) or nm == "gibbs_sweeps") if not args.breakdown else (1 << len(names)) - 1
    # The autoencoder half of every step is replayed from a captured hipGraph (one graph launch instead of ~120 kernel
    # launches); on every 10th step the GRBM quasi-NLL update runs eagerly behind it.  --eager disables the graph.
    # (several GPUs: two graphs per step with the eager RCCL all-reduce of the flat gradient buffer between them)
    model.use_graph = not args.eager
    for kv in args.wrapper:  # A/B runs: ModelWrapper attributes (prepare_decoder=0, defer_mmd_join=1, ...)
        name, _, value = kv.partition("=")
        if not hasattr(model, name):
            raise SystemExit(f"--wrapper: ModelWrapper has no attribute {name!r}")
        setattr(model, name, {"none": None}.get(value.lower(), int(value) if value.lstrip("-").isdigit() else value))
    step_idx = 0

    def run(k):
        nonlocal step_idx
        for _ in range(k):
            model.step(batches[step_idx % pool], epoch=0)
            step_idx += 1

    if model.use_graph:
        run(4 + model.N_GRAPHS)  # 3 eager steps + the captures + first replays: not part of the W warm-up steps
    run(args.warmup)
    torch.cuda.synchronize()
    L.dvg_prof_reset()
    if not model.use_graph:
        L.dvg_prof_enable(mask)  # eager: per-kernel HIP events over the timed region itself
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dp.barrier()
    elapsed = time.perf_counter() - t0
    L.dvg_prof_enable(0)
    elapsed = dp.max_over_ranks(elapsed)
    # What the timed region computed: the last timed step's losses (device scalars the step left in model.last) and the
    # parameters it produced.  A run whose numbers are not finite is not a measurement.
    last_mse, last_mmd = float(model.last["mse"]), float(model.last["mmd"])
    params_finite = bool(torch.isfinite(model._dvae_optimizer.flat).all()) and bool(torch.isfinite(model._grbm_optimizer.flat).all())
    workload_losses = {"mse": last_mse, "mmd": last_mmd, "mse+mmd": last_mse + last_mmd,
                       "nll": float(model.last["nll"]) if "nll" in model.last else None,
                       "params_finite": params_finite, "step": step_idx - 1,
                       "note": "losses of the LAST timed step and finiteness of every parameter after the timed region"}
    if not (params_finite and all(v == v and abs(v) != float("inf") for v in (last_mse, last_mmd))):
        raise SystemExit(f"bench: rank {dp.rank}: the timed steps produced non-finite numbers: {workload_losses}")
    per_rank = dp.gather_objects({"rank": dp.rank, "dist": dp.describe(), "mse": last_mse, "mmd": last_mmd})
    prof_steps = args.steps
    import ctypes

    def query_kernels():
        res = {}
        for i, nm in enumerate(names):
            ms, cnt, work, share = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            L.dvg_prof_query(i, ctypes.byref(ms), ctypes.byref(cnt))
            L.dvg_prof_query_work(i, ctypes.byref(work))
            L.dvg_prof_query_share(i, ctypes.byref(share))
            if cnt.value:
                # share_ms: duration x the share of the chip's CUs the launch's grid was sized for (1 except for the
                # CU-budgeted Winograd grids, which are handed 128-224 CUs because another kernel holds the rest)
                res[nm] = {"total_ms": ms.value, "launches": cnt.value, "work": work.value, "share_ms": share.value}
        return res

    passes = []
    if model.use_graph:
        # hipGraph replays cannot carry per-kernel event records, so the per-kernel HIP-event timing behind
        # `roofline` comes from eager passes of the SAME steps right after the timed region (same kernels, same
        # shapes, same streams).  THREE passes, the per-kernel MEDIAN is reported and the spread of the headline
        # fraction beside it (`frac_passes`): how the host paces an eager step moves what runs beside what -- one pass
        # gave 0.28 .. 0.34 for the same 9.04 ms step in round 4.
        model.use_graph = False
        prof_steps = max(10, min(args.steps, 20))
        for _ in range(3):
            L.dvg_prof_reset()
            L.dvg_prof_enable(mask)
            run(prof_steps)
            torch.cuda.synchronize()
            L.dvg_prof_enable(0)
            passes.append(query_kernels())
        model.use_graph = True
    else:
        passes.append(query_kernels())
    per_kernel = {}
    for nm in passes[0]:
        rows = sorted((p_[nm] for p_ in passes if nm in p_), key=lambda r_: r_["total_ms"])
        per_kernel[nm] = dict(rows[len(rows) // 2], total_ms_passes=[r_["total_ms"] for r_ in rows])
    if dp.rank == 0:
        # dominant kernel = the GEMM kernel (one template instantiation = one rocprof kernel name) with the
        # largest total time; achieved = its executed FLOPs (2*rows*Cin*Cout*taps per launch -- the folded-upsample
        # layers are credited their 4 or 16 folded taps, not the 9 they replace -- summed by the library over the
        # timed launches) / its HIP-event time over the timed region.
        # The MMD pair kernel serves +-1 spin rows ("mmd_pm1") at two matrix-pipe rates at once: the Gram of every
        # visited pair on the int8 MFMA (2x the bf16 rate) and the gradient GEMM as 2 (large problems) or 3 bf16 terms per
        # weight on the bf16 MFMA.  It is priced per type: the library reports its work in bf16-equivalent FLOPs (int8 FLOPs x 0.5 +
        # bf16 FLOPs), i.e. frac = (int8 FLOPs / 5 PF + bf16 FLOPs / 2.5 PF) / measured time; `pricing` spells the two
        # parts out.  General rows run on the f32 MFMA ("mmd_main").  A candidate whose rate exceeds its peak is a
        # mislabelled launch and is dropped.
        lib_hash = L.dvg_source_hash().decode()
        # --precision f32x3: the forward / data-gradient launches run their float32 operands as three bf16 pieces,
        # i.e. they EXECUTE six bf16 products per algorithmic multiply-add: priced as 6 x the algorithmic FLOPs against the
        # bf16 peak (the library's work counter holds the algorithmic count; `algorithmic_f32_tflops` keeps it visible)
        split_gemm = lambda nm: args.precision == "f32x3" and nm.startswith("conv_igemm_kernel")  # noqa: E731  (every forward / data-gradient launch; the weight-space products have their own id)
        for k, v in per_kernel.items():
            if split_gemm(k):
                v["algorithmic_work"] = v["work"]
                v["work"] = 6.0 * v["work"]
        bf16_gemm = lambda nm: nm == "mmd_pm1" or split_gemm(nm) or (args.precision == "bf16" and nm.startswith("conv_igemm_kernel"))  # noqa: E731
        peak_of = lambda nm: PEAK_BF16_MFMA_TFLOPS if bf16_gemm(nm) else PEAK_F32_MFMA_TFLOPS  # noqa: E731
        cands = {k: v for k, v in per_kernel.items() if is_gemm(k) and v["work"] > 0
                 and v["work"] / (v["total_ms"] * 1e-3) / 1e12 <= peak_of(k)}

        def entry(k):
            v = cands[k]
            ach = v["work"] / (v["total_ms"] * 1e-3) / 1e12
            e = {"kernel": k, "bound": "mfma", "achieved": ach, "peak": peak_of(k), "unit": "TFLOP/s",
                 "frac": ach / peak_of(k), "avg_launch_us": v["total_ms"] * 1e3 / v["launches"], "launches": v["launches"],
                 "gflop_per_launch": v["work"] / v["launches"] / 1e9, "ms_per_step": v["total_ms"] / prof_steps}
            if len(v.get("total_ms_passes", ())) > 1:  # (slowest .. fastest pass; `frac` is the median pass)
                e["frac_passes"] = [v["work"] / (t * 1e-3) / 1e12 / peak_of(k) for t in sorted(v["total_ms_passes"], reverse=True)]
            if v.get("share_ms", 0.0) > 0.0 and v["share_ms"] < 0.999 * v["total_ms"]:
                # a persistent whole-CU grid sized to a CU budget (conv_wino*.hip): `frac` above prices it against the WHOLE
                # chip's peak although the launch was given part of it; this is the same work against the CUs it was given
                e["cu_share"] = v["share_ms"] / v["total_ms"]
                e["frac_of_cu_budget"] = v["work"] / (v["share_ms"] * 1e-3) / 1e12 / peak_of(k)
            if "algorithmic_work" in v:
                e["unit"] = "TFLOP/s (bf16 products executed: 6 per algorithmic float32 multiply-add)"
                e["algorithmic_f32_tflops"] = v["algorithmic_work"] / (v["total_ms"] * 1e-3) / 1e12
            e["traffic"], e["traffic_source"] = pmc_traffic(k, args.config, lib_hash)
            e["mfma_busy_pmc"] = pmc_mfma_busy(k, args.config, lib_hash) if args.precision == "f32" else None
            if args.precision == "f32":  # (one profiler id, several template instantiations: each one's PMC figure)
                e["mfma_busy_pmc_instantiations"] = pmc_mfma_busy_all(k, args.config, lib_hash)
            if k == "mmd_pm1":
                i8, b16, terms = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
                _lib.check(L.dvg_mmd_spin_flops(cfg["B"] * cfg["R"], cfg["C"], cfg["n"], ctypes.byref(i8), ctypes.byref(b16),
                                                ctypes.byref(terms)), "dvg_mmd_spin_flops")
                nx, ny, d = cfg["B"] * cfg["R"], cfg["C"], cfg["n"]
                e["unit"] = "TFLOP/s (bf16-equivalent: int8 FLOPs x 0.5 + bf16 FLOPs)"
                e["pricing"] = {"int8_gram_tflop_per_launch": i8.value / 1e12, "int8_peak_tflops": 2 * PEAK_BF16_MFMA_TFLOPS,
                                "bf16_gradient_tflop_per_launch": b16.value / 1e12, "bf16_terms": terms.value,
                                "bf16_peak_tflops": PEAK_BF16_MFMA_TFLOPS,
                                "time_at_peak_us": (i8.value / (2 * PEAK_BF16_MFMA_TFLOPS) + b16.value / PEAK_BF16_MFMA_TFLOPS) / 1e6,
                                "algorithmic_tflop_per_launch": (2.0 * (nx + ny) ** 2 * d + 2.0 * nx * (nx + ny) * d) / 1e12}
            return e

        dom = max(cands, key=lambda k: cands[k]["total_ms"])
        roofline = entry(dom)
        roofline["timing"] = ("HIP events over the timed region" if args.eager else
                              f"HIP events over three eager passes of {prof_steps} steps right after the timed region, per-kernel "
                              "median (graph replays cannot carry per-kernel events); in-situ durations: the step runs the "
                              "sampler / MMD / weight-gradient chains beside the critical chain on other streams")
        roofline["kernels_hash"] = lib_hash
        conv = {k: v for k, v in cands.items() if k.startswith("conv_")}
        if conv:  # the convolution GEMM beside it (north star: >= 40 % MFMA utilisation on the encoder/decoder GEMMs)
            roofline["conv"] = entry(max(conv, key=lambda k: conv[k]["total_ms"]))
            tw, tt = sum(v.get("algorithmic_work", v["work"]) for v in conv.values()), sum(v["total_ms"] for v in conv.values())
            # `reference_equivalent`: what those kernels DELIVER in the reference's own operation count (the
            # layer-by-layer 9-tap network, net_flops_per_image) -- the folded upsample, the composed first decoder layers,
            # position-major tiles and the Winograd forms execute 0.3-0.45 of it; a rate above the f32 MFMA peak here is
            # algebra, not a mislabelled launch (the executed rate `frac` is the utilisation figure)
            ref_gflop = net_flops_per_image(cfg["n"], cfg["R"]) * cfg["B"] / 1e9
            roofline["conv_all"] = {"tflops": tw / (tt * 1e-3) / 1e12, "note": "EXECUTED float32 FLOPs of all convolution GEMM kernels (Winograd launches: their 36 position GEMMs per 4x4 tile -- F(4x4,3x3) -- or 16 / 9 per 2x2 quad / their summed time; frac against the f32 MFMA peak (bf16 peak in --precision bf16)",
                                    "reference_equivalent": {"gflop_per_step": ref_gflop, "tflops": ref_gflop * prof_steps / (tt * 1e-3) / 1e3,
                                                             "executed_over_reference": tw / prof_steps / 1e9 / ref_gflop},
                                    "frac": tw / (tt * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TFLOPS if args.precision == "bf16" else PEAK_F32_MFMA_TFLOPS),
                                    "ms_per_step": tt / prof_steps, "gflop_per_step": tw / prof_steps / 1e9}
            ts = sum(v.get("share_ms", v["total_ms"]) for v in conv.values())
            if 0.0 < ts < 0.999 * tt:  # (see frac_of_cu_budget above: the Winograd launches are sized to part of the chip)
                roofline["conv_all"]["frac_of_cu_budget"] = tw / (ts * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TFLOPS if args.precision == "bf16" else PEAK_F32_MFMA_TFLOPS)
                roofline["conv_all"]["cu_share"] = ts / tt
        if "mmd_pm1" in cands and dom != "mmd_pm1":
            roofline["mmd_pair"] = entry("mmd_pm1")
        roofline["sampler"] = sampler_roofline(per_kernel, cfg, model.sampler.plan, prof_steps, args.config, lib_hash)
        roofline["all_gemm_kernels"] = {k: {"tflops": v["work"] / (v["total_ms"] * 1e-3) / 1e12,
                                            "ms_per_step": v["total_ms"] / prof_steps,
                                            "avg_launch_us": v["total_ms"] * 1e3 / v["launches"]} for k, v in cands.items()}
        ips = args.gpus * cfg["B"] * args.steps / elapsed
        out = {
            "metric": baseline_metric(), "metric_id": "dvae_grbm_train_step_images_per_s", "value": ips,
            "unit": "images/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16": "bf16 GEMM inputs, f32 accumulate (weight gradients f32)",
                      "f32x3": "f32 (operands of the large forward / data-gradient GEMM launches as three bf16 pieces, six piece "
                               "products per k-step on the bf16 MFMA, f32 accumulate; everything else f32)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {cfg['desc']}", "global_batch": cfg["B"] * args.gpus,
                       "n_latents": cfg["n"], "n_replicas": cfg["R"], "num_reads_per_gpu": cfg["C"],
                       "gibbs_sweeps": cfg["sweeps"], "parallelism": f"dp{args.gpus}",
                       "launch": ("hipGraph replay of the autoencoder half (GRBM update of every 10th step eager behind it)"
                                  if model.use_graph else "eager"),
                       # (what the wrapper actually did over the whole process: replays of the captured step / eager steps /
                       # whether a capture failed and the run fell back to eager launches)
                       "launch_counts": {"graph_replays": int(model._replays), "eager_steps": int(model._eager_steps),
                                         "capture_failed": bool(model._graph_failed),
                                         "replay_call_host_us": round(1e6 * model._replay_host_s / max(1, int(model._replays)), 1)},
                       "net_gflop_per_step": net_flops_per_image(cfg["n"], cfg["R"]) * cfg["B"] / 1e9},
            "workload_losses": workload_losses,
            # what the process group looked like from inside (not what --gpus claimed): backend, world size RCCL reports,
            # every rank's device
            "dist": dict(dp.describe(), ranks=[{"rank": r["rank"], "device": r["dist"].get("device"),
                                                "device_name": r["dist"].get("device_name"),
                                                "world_size_seen": r["dist"].get("world_size_seen"),
                                                "mse": r["mse"], "mmd": r["mmd"]} for r in per_rank]),
            "roofline": roofline,
        }
        if args.gpus == 1 and not args.no_cpu_baseline:
            try:  # (a failed or timed-out child must not cost the line its GPU measurement)
                out["cpu_baseline"] = cpu_baseline(args.config)
            except Exception as exc:
                out["cpu_baseline"] = {"error": repr(exc)}
            out["loss_parity"] = loss_parity()
        if args.parity and args.gpus == 1:
            out["loss_parity"] = dict(loss_parity(args.precision if args.precision != "f32" else None),
                                      note="12 fixture steps in this arithmetic mode (every forward / data-gradient launch)")
        if args.gpus == 1 and not args.child and not args.no_cpu_baseline and args.precision == "f32":
            out["f32x3"] = split3_run(args)
            out["bf16_inputs"] = bf16_inputs_run(args)
            if args.config == "c3":
                out["extra"] = {c: extra_config_run(args, c) for c in ("c2", "c1")}
                # BASELINE.json configs[0] IS "CPU PyTorch reference": c1 at its own batch on the host cores -- the one
                # pair of this line where the CPU and the GPU ran the same workload at the same size
                if "error" not in out["extra"]["c1"]:
                    try:
                        out["extra"]["c1"]["cpu_baseline"] = cpu_baseline("c1", seconds=8.0)
                        out["extra"]["c1"]["gpu_over_cpu"] = out["extra"]["c1"]["value"] / out["extra"]["c1"]["cpu_baseline"]["value"]
                    except Exception as exc:
                        out["extra"]["c1"]["cpu_baseline"] = {"error": repr(exc)}
                out["layerwise_first_layers"] = layerwise_run(args)
        if args.breakdown:
            with open(args.breakdown, "w") as f:
                json.dump({"ms_per_step": elapsed / args.steps * 1e3, "profiled_steps": prof_steps, "kernels": per_kernel}, f, indent=1)
        line = json.dumps(out)
    else:
        line = None
    # The JSON line is the LAST thing this process prints: RCCL writes a version banner through C stdio, which is
    # fully buffered when stdout is a pipe or file and would otherwise be flushed at exit, i.e. AFTER the line.
    dp.shutdown()
    os.unlink(tmp.name)
    sys.stdout.flush()
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
