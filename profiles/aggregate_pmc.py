"""Aggregate two rocprofv3 PMC passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE, each with --kernel-trace) of
`bench.py --eager` into HBM bytes per launch per kernel.

    python profiles/aggregate_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> out.json [kernel_trace.csv]

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters
report KiB; FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE is exact.
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


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f_kib, n = fetch.get(k, (0.0, 0))
        w_kib, n2 = write.get(k, (0.0, 0))
        out[k] = {"launches": max(n, n2), "FETCH_SIZE_KiB_avg": f_kib, "WRITE_SIZE_KiB_avg": w_kib,
                  "hbm_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0}
    if len(sys.argv) > 4:
        # optional: the kernel_trace.csv of one of the passes -> average dispatch duration (serialised under PMC collection,
        # i.e. each kernel alone) and the HBM rate it implies against the 8 TB/s peak
        dur = defaultdict(lambda: [0.0, 0])
        with open(sys.argv[4]) as f:
            for r in csv.DictReader(f):
                d = dur[r["Kernel_Name"]]
                d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                d[1] += 1
        for k, rec in out.items():
            if dur[k][1]:
                us = dur[k][0] / dur[k][1] / 1e3
                rec["avg_us_alone"] = us
                rec["hbm_GBps"] = rec["hbm_bytes_per_launch"] / (us * 1e-6) / 1e9
                rec["hbm_frac_of_8TBps"] = rec["hbm_GBps"] / 8000.0
    json.dump({"kernels_hash": kernels_hash(), "kernels": out}, open(sys.argv[3], "w"), indent=1)
    print(f"{len(out)} kernels -> {sys.argv[3]}")


if __name__ == "__main__":
    main()
