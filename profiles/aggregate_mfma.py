"""Aggregate one rocprofv3 PMC pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, with --kernel-trace) of
`bench.py --eager` into the matrix-pipe utilisation per kernel.

    python profiles/aggregate_mfma.py <counter_collection.csv> out.json

Units (MI355X_MICROARCH.md, 'Per-instruction cycle constants'): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over
the SIMDs (64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is summed over the 8 XCDs,
so the dispatch lasted GRBM_GUI_ACTIVE / 8 cycles at the clock it actually ran at.  Utilisation = busy cycles /
(1024 SIMDs x dispatch cycles): independent of the DVFS clock.
"""
import csv
import json
import os
import sys
from collections import defaultdict


def kernels_hash():
    """dvg_source_hash() of the library in this tree = the build the profile was measured on (bench.py ignores a
    profile whose hash differs from the library it runs)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import image_generation_amd  # noqa: F401
    from image_generation_amd import _lib

    return _lib.lib().dvg_source_hash().decode()

N_SIMD = 256 * 4


def main():
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]][r["Counter_Name"]] += 1
    out = {}
    for k, c in acc.items():
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        n = cnt[k].get("GRBM_GUI_ACTIVE", 0)
        if busy <= 0 or act <= 0 or not n:
            continue
        out[k] = {"launches": n, "mfma_busy_cycles_avg": busy / n, "dispatch_cycles_avg": act / 8 / n,
                  "mfma_busy_frac": busy / (N_SIMD * act / 8)}
    with open(sys.argv[2], "w") as f:
        ranked = dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_avg"] * kv[1]["launches"]))
        json.dump({"kernels_hash": kernels_hash(), "kernels": ranked}, f, indent=1)
    print(f"{len(out)} MFMA kernels -> {sys.argv[2]}")


if __name__ == "__main__":
    main()
