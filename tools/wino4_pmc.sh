#!/bin/bash
# PMC counters of the F(4x4,3x3) kernel alone (tools/wino4_bench.py), run through gpurun from the repo root:
#   gpurun -- 'bash tools/wino4_pmc.sh'
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/wino4_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=${1:-wino4_bench}
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/a_$T -- python3 $ROOT/tools/$T.py > $OUT/a_$T.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b_$T -- python3 $ROOT/tools/$T.py > $OUT/b_$T.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/wino4_pmc/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-44:] + " grid " + r.get("Grid_Size", "?")
        if "conv_wino" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    print(f.split("/")[2])
    for k, c in sorted(acc.items()):
        print(" ", k)
        for n, v in sorted(c.items()):
            print(f"     {n:32s} {v / cnt[(k, n)]:16.0f} per launch")
PY
