"""Instruction counts of the block-Gibbs draw from a rocprofv3 PMC pass of tools/gibbs_bench.py (the draw alone):

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
        --output-format csv -d DIR -- python3 tools/gibbs_bench.py N CHAINS SWEEPS default
    python profiles/aggregate_insts.py DIR/*/*counter_collection.csv out.json N CHAINS SWEEPS

What bench.py makes of it (`roofline.sampler.issue_bound`): the draw is bound by instruction ISSUE -- one wave per SIMD
issues in program order, one vector / LDS / scalar instruction per 4 cycles at best (MI355X_MICROARCH.md, row
"vector-instruction ISSUE cost": one wave's stream on one SIMD) -- so its floor is  (wave-instructions per wave) x 4
cycles / clock, or, where several waves share a SIMD, (VALU wave-instructions per SIMD) x 2 cycles (a wave64 VALU
instruction occupies the SIMD-32 for two).  The fraction printed is that floor / the measured draw time: what is left is
latency the kernel does not hide; everything else can only be bought by issuing fewer instructions."""
import csv
import json
import sys
from collections import defaultdict

from aggregate_pmc import kernels_hash


def main():
    path, out_path = sys.argv[1], sys.argv[2]
    n, chains, sweeps = (int(v) for v in sys.argv[3:6])
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    geom = {}
    dur = defaultdict(lambda: [0.0, 0])
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if "gibbs" not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
            geom[k] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["LDS_Block_Size"]), int(r["VGPR_Count"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                dur[k][1] += 1
    out = {}
    for k, c in acc.items():
        g = {m: c[m] / cnt[k][m] for m in c}
        grid, wg, lds, vgpr = geom[k]
        waves, groups = grid // 64, grid // wg
        cus = min(256, groups)
        waves_per_simd = (wg // 64) / 4.0 * max(1.0, groups / 256.0)
        total = sum(g.get(m, 0.0) for m in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"))
        out[k] = {
            "launches": cnt[k]["SQ_INSTS_VALU"], "grid_threads": grid, "workgroup_threads": wg, "workgroups": groups, "waves": waves,
            "cus_used": cus, "waves_per_simd": waves_per_simd, "lds_bytes_per_workgroup": lds, "vgprs": vgpr,
            "insts_per_launch": {m[len("SQ_INSTS_"):].lower(): g.get(m, 0.0) for m in g if m.startswith("SQ_INSTS_")},
            "insts_per_wave_per_sweep": total / waves / sweeps,
            "valu_per_spin_update": g.get("SQ_INSTS_VALU", 0.0) * 64.0 / (float(n) * chains * sweeps),
            "lds_per_spin_update": g.get("SQ_INSTS_LDS", 0.0) * 64.0 / (float(n) * chains * sweeps),
            "avg_us_alone_under_pmc": dur[k][0] / max(1, dur[k][1]) / 1e3,
        }
    json.dump({"kernels_hash": kernels_hash(), "n": n, "chains": chains, "sweeps": sweeps, "kernels": out}, open(out_path, "w"), indent=1)
    print(f"{len(out)} kernels -> {out_path}")


if __name__ == "__main__":
    main()
