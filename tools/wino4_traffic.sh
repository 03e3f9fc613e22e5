#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes: MI355X_MICROARCH.md) of the F(4x4,3x3) and F(2x2,3x3) kernels alone,
# one launch shape per pass:   gpurun -- 'bash tools/wino4_traffic.sh'
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/wino4_traffic; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for S in "L1 fwd" "L2 fwd" "L3 fwd" "L3 dgrad"; do
  T=$(echo $S | tr ' ' '_')
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${C}_$T -- python3 $ROOT/tools/wino4_bench.py 4096 "$S" > $OUT/${C}_$T.log 2>&1
  done
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/wino4_traffic/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:] + " grid " + r.get("Grid_Size", "?")
        if "conv_wino" in k or "conv_igemm" in k:
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for (k, n), v in sorted(acc.items()):
        kib = v / cnt[(k, n)]
        print(f"{f.split('/')[2]:24s} {k:60s} {n:11s} {kib / 1024 * (2 if n == 'FETCH_SIZE' else 1):9.1f} MiB per launch" + (" (x2: gfx950 correction)" if n == "FETCH_SIZE" else ""))
PY
