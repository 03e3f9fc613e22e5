#!/bin/bash
# Regenerates everything under profiles/ on a GPU box (run from the repo root through gpurun; outputs land in
# gpurun_out/profiles_new/, to be copied into profiles/ and committed):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/make_profiles.sh r01'
set -u
R=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_new
mkdir -p $OUT
last() { grep '^{"metric"' "$1" | tail -1 > "$2"; }
# 1. bench lines (c2 carries the per-kernel HIP-event breakdown and the CPU baseline; c1 / c3 for the record)
python bench.py --config c2 --breakdown $OUT/${R}_hip_event_breakdown_c2.json > $OUT/c2.log 2>&1; last $OUT/c2.log $OUT/${R}_bench_c2.json
python bench.py --config c1 > $OUT/c1.log 2>&1; last $OUT/c1.log $OUT/${R}_bench_c1.json
python bench.py --config c3 --no-cpu-baseline --steps 10 > $OUT/c3.log 2>&1; last $OUT/c3.log $OUT/${R}_bench_c3.json
python bench.py --config c5 --no-cpu-baseline --steps 20 > $OUT/c5.log 2>&1; last $OUT/c5.log $OUT/${R}_bench_c5.json
python bench.py --config c3 --precision bf16 --no-cpu-baseline --steps 10 > $OUT/c3b.log 2>&1; last $OUT/c3b.log $OUT/${R}_bench_c3_bf16_inputs.json
# 2. rocprofv3 kernel stats of the same c2 command
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --config c2 --no-cpu-baseline > $OUT/c2_rocprof.log 2>&1
last $OUT/c2_rocprof.log $OUT/${R}_bench_c2_under_rocprof.json
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${R}_rocprofv3_kernel_stats_c2.csv
# 3. HBM traffic: separate PMC passes (eager launches so every kernel is its own dispatch)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -- python3 $ROOT/bench.py --config c2 --eager --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -- python3 $ROOT/bench.py --config c2 --eager --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_m -- python3 $ROOT/bench.py --config c2 --eager --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f3 -- python3 $ROOT/bench.py --config c3 --eager --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w3 -- python3 $ROOT/bench.py --config c3 --eager --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_m3 -- python3 $ROOT/bench.py --config c3 --eager --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
cd $ROOT
python profiles/aggregate_mfma.py $(ls $OUT/pmc_m3/*/*counter_collection.csv | head -1) $OUT/${R}_pmc_mfma_busy_c3.json
python profiles/aggregate_mfma.py $(ls $OUT/pmc_m/*/*counter_collection.csv | head -1) $OUT/${R}_pmc_mfma_busy_c2.json
python profiles/aggregate_pmc.py $(ls $OUT/pmc_f/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_w/*/*counter_collection.csv | head -1) $OUT/${R}_pmc_traffic_c2.json $(ls $OUT/pmc_w/*/*kernel_trace.csv | head -1)
python profiles/aggregate_pmc.py $(ls $OUT/pmc_f3/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_w3/*/*counter_collection.csv | head -1) $OUT/${R}_pmc_traffic_c3.json $(ls $OUT/pmc_w3/*/*kernel_trace.csv | head -1)
rm -rf $OUT/stats $OUT/pmc_f $OUT/pmc_w $OUT/pmc_m $OUT/pmc_m3 $OUT/pmc_f3 $OUT/pmc_w3 $OUT/*.log
ls -la $OUT
