#!/bin/bash
# Instruction mix per kernel of one eager c3 step (VALU / SALU / LDS / MFMA wave-instructions per launch, MFMA-busy): where
# does a matrix kernel spend issue slots on things that are not matrix work?   gpurun -- 'bash tools/step_pmc_mix.sh [config]'
set -u
C=${1:-c3}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/step_mix; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --config $C --eager --no-cpu-baseline --child --steps 3 --warmup 2 > $OUT/a.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/step_mix/a/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void dvg::", "")[:58]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
rows = []
for k, c in acc.items():
    n = cnt[(k, "SQ_INSTS_VALU")] or 1
    g = {m: c[m] / (cnt[(k, m)] or 1) for m in c}
    rows.append((g.get("GRBM_GUI_ACTIVE", 0) * n, k, n, g))
print(f"{'kernel':58s} {'launches':>8s} {'Mcyc/launch':>11s} {'VALU/MFMA':>9s} {'SALU/MFMA':>9s} {'LDS/MFMA':>8s} {'mfma busy':>9s}")
for tot, k, n, g in sorted(rows, reverse=True)[:28]:
    mf = g.get("SQ_INSTS_MFMA", 0)
    cyc = g.get("GRBM_GUI_ACTIVE", 0) / 8
    busy = g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc) if cyc else 0
    r = lambda x: (g.get(x, 0) / mf) if mf else float("nan")
    print(f"{k:58s} {n:8d} {cyc/1e6:11.3f} {r('SQ_INSTS_VALU'):9.2f} {r('SQ_INSTS_SALU'):9.2f} {r('SQ_INSTS_LDS'):8.2f} {busy:9.2f}")
PY
