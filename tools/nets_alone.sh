#!/bin/bash
# Per-kernel durations of the encoder and of the decoder run BY THEMSELVES at c3's batch (rocprofv3 --kernel-trace --stats):
# once as the training step runs them (weight-gradient chain on the library's side stream) and once with that stream off
# (every kernel alone on the chip).  An in-step duration says little about a kernel: the step is work-conserving.
#   bash tools/nets_alone.sh > gpurun_out/nets_alone.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/nets_alone; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for NET in encoder decoder; do
  for SS in 1 0; do
    rm -rf $OUT/tr
    if [ $SS = 0 ]; then export OPTS=side_stream=0; else unset OPTS; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python3 $ROOT/tools/${NET}_alone.py > $OUT/log.txt 2>&1
    echo "== $NET alone, c3 batch, side stream $([ $SS = 1 ] && echo on || echo off): calls, average us, minimum us"
    python3 - "$(ls $OUT/tr/*/*kernel_stats.csv | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dvg::" in r["Name"]:
        print("  %-78s %4s %8.1f %8.1f" % (r["Name"].replace("void ", "")[:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  done
done
rm -rf $OUT
