#!/bin/bash
# Regenerates everything under profiles/ on a GPU box (run from the repo root through gpurun; outputs land in
# gpurun_out/profiles_new/, to be copied into profiles/ and committed):
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/make_profiles.sh r03'
# Every JSON written by the aggregators carries kernels_hash = dvg_source_hash() of the library measured; bench.py drops
# profiles whose hash differs from the library it runs.
set -u
R=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_new
mkdir -p $OUT
last() { grep '^{"metric"' "$1" | tail -1 > "$2"; }
# `bash tools/make_profiles.sh r03 bench`: only the bench lines of step 3 (the PMC files already in profiles/ must be of
# this build of the kernels: bench.py checks their hash)
if [ "${2:-all}" != "bench" ]; then
# 1. rocprofv3 kernel stats of the headline command (c3) and of c2
cd /tmp && export TMPDIR=/tmp
for C in c3 c2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$C -- python3 $ROOT/bench.py --config $C --no-cpu-baseline > $OUT/${C}_rocprof.log 2>&1
  last $OUT/${C}_rocprof.log $OUT/${R}_bench_${C}_under_rocprof.json
  cp $(ls $OUT/stats_$C/*/*kernel_stats.csv | head -1) $OUT/${R}_rocprofv3_kernel_stats_$C.csv
done
# 2. HBM traffic / MFMA-busy: separate PMC passes (eager launches so every kernel is its own dispatch)
for C in c3 c2; do
  if [ $C = c3 ]; then ST="--steps 3 --warmup 2"; else ST="--steps 10 --warmup 3"; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f_$C -- python3 $ROOT/bench.py --config $C --eager --no-cpu-baseline $ST > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w_$C -- python3 $ROOT/bench.py --config $C --eager --no-cpu-baseline $ST > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_m_$C -- python3 $ROOT/bench.py --config $C --eager --no-cpu-baseline $ST > /dev/null 2>&1
done
# 2b. instruction counts of the sampler's draw alone (the issue bound of roofline.sampler): c3, c5 slice, c2
for SPEC in "c3 512 256 200 zephyr" "c5 1024 2048 50 zephyr" "c2 128 256 50 pegasus"; do
  set -- $SPEC
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_g_$1 -- python3 $ROOT/tools/gibbs_bench.py $2 $3 $4 default $5 > /dev/null 2>&1
  (cd $ROOT/profiles && python aggregate_insts.py $(ls $OUT/pmc_g_$1/*/*counter_collection.csv | head -1) $OUT/${R}_pmc_gibbs_insts_$1.json $2 $3 $4)
done
set -- $R
cd $ROOT
for C in c3 c2; do
  python profiles/aggregate_mfma.py $(ls $OUT/pmc_m_$C/*/*counter_collection.csv | head -1) $OUT/${R}_pmc_mfma_busy_$C.json
  python profiles/aggregate_pmc.py $(ls $OUT/pmc_f_$C/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_w_$C/*/*counter_collection.csv | head -1) $OUT/${R}_pmc_traffic_$C.json $(ls $OUT/pmc_w_$C/*/*kernel_trace.csv | head -1)
done
cp $OUT/${R}_pmc_*.json $ROOT/profiles/
fi
# 3. bench lines LAST (they read the PMC files of this build from profiles/)
cd $ROOT
T0=$SECONDS; python bench.py --breakdown $OUT/${R}_hip_event_breakdown_c3.json > $OUT/c3.log 2>&1; echo "default bench.py (all child runs, CPU baseline, parity): $((SECONDS - T0)) s wall" > $OUT/${R}_bench_default_wall_seconds.txt; last $OUT/c3.log $OUT/${R}_bench_c3.json; tail -5 $OUT/c3.log > $OUT/c3_tail.txt
python bench.py --config c2 --no-cpu-baseline --steps 30 --breakdown $OUT/${R}_hip_event_breakdown_c2.json > $OUT/c2.log 2>&1; last $OUT/c2.log $OUT/${R}_bench_c2.json
python bench.py --config c5 --no-cpu-baseline --steps 20 > $OUT/c5.log 2>&1; last $OUT/c5.log $OUT/${R}_bench_c5.json
python bench.py --config c3 --precision bf16 --no-cpu-baseline --steps 10 > $OUT/c3b.log 2>&1; last $OUT/c3b.log $OUT/${R}_bench_c3_bf16_inputs.json
python bench.py --config c3 --precision f32x3 --no-cpu-baseline --parity --steps 20 > $OUT/c3x.log 2>&1; last $OUT/c3x.log $OUT/${R}_bench_c3_f32x3.json
python tools/mmd_accuracy.py 2>&1 | grep -v amdgpu > $OUT/${R}_mmd_accuracy_c3.txt
python tools/mmd_bench.py 2>&1 | grep -v amdgpu > $OUT/${R}_mmd_kernels_c3.txt
# the c5 slice's MMD (2048 + 2048 rows, d = 1024): the 128-row-block kernel in eight feature slices, and the 32-row kernel it replaced
(python tools/mmd_bench.py 2048 2048 1024; python tools/mmd_bench.py 2048 2048 1024 --w128 0) 2>&1 | grep -v amdgpu > $OUT/${R}_mmd_kernels_c5.txt
python tools/wino4_dec_bench.py 2>&1 | grep -v amdgpu > $OUT/${R}_wino4_decoder_alone_c3.txt
python tools/igemm_ab.py 2>&1 | grep -v amdgpu > $OUT/${R}_igemm_staging_ab.txt
python tools/igemm_modes.py 2>&1 | grep -v amdgpu > $OUT/${R}_igemm_operand_modes.txt
python tools/gibbs_bench.py 2>&1 | grep -v amdgpu > $OUT/${R}_gibbs_draw_alone_c3.txt
python tools/gibbs_bench.py 1024 2048 50 2>&1 | grep -v amdgpu > $OUT/${R}_gibbs_draw_alone_c5.txt
# the driver's multi-GPU command shape, with the one rank this box has (a forced single-rank RCCL group): the line's `dist`
# object shows what the process group looked like from inside
HSA_ENABLE_IPC_MODE_LEGACY=0 DVG_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --child > $OUT/tr1.log 2>&1; last $OUT/tr1.log $OUT/${R}_bench_c3_torchrun_nproc1_forced_dist.json
# kernel timeline of one REPLAYED c3 step (index -35: behind the 10 timed steps bench.py runs 3 x 10 EAGER steps for its per-kernel
# HIP-event timing; round 6 found the committed timelines of rounds 4-5 were of those -- for c3 the same picture, for c2 not)
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $ROOT/bench.py --no-cpu-baseline --child --steps 10 --warmup 3 > /dev/null 2>&1; cd $ROOT
python tools/trace_step.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) -35 | cut -c1-120 > $OUT/${R}_timeline_c3_step.txt; rm -rf $OUT/tr
python tools/wgrad_ab.py 2>&1 | grep -v amdgpu > $OUT/${R}_wgrad_staging_ab.txt
python tools/wino_bench.py 2>&1 | grep -v amdgpu > $OUT/${R}_wino_alone_c3.txt
python tools/wino_wgrad_bench.py 2>&1 | grep -v amdgpu > $OUT/${R}_wino_wgrad_alone_c3.txt
# the F(4x4,3x3) kernels alone, beside the F(2x2,3x3) and direct kernels on the same launches
python tools/wino4_bench.py 2>&1 | grep -v amdgpu > $OUT/${R}_wino4_alone_c3.txt
python tools/wino4_wgrad_bench.py 2>&1 | grep -v amdgpu > $OUT/${R}_wino4_wgrad_alone_c3.txt
python tools/wino4_sweep.py 2>&1 | grep -v amdgpu > $OUT/${R}_wino4_time_per_chunk.txt
# every F(4x4) instantiation against float64, with the F(2x2) and direct kernels' distances beside it (the kernel tests' own table)
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "f4x4" -s 2>&1 | grep -E "^\.?F\(4x4\)|passed|failed" | sed 's/^\.//' > $OUT/${R}_wino4_accuracy_vs_float64.txt
# determinism: two processes, 300 graph-replayed c3 steps each, losses printed to the last digit (dynamic tile deal, pair
# exchange and fixed-order slab sums included)
(python tools/soak.py c3 300 2>&1 | tail -3; python tools/soak.py c3 300 2>&1 | tail -3) | grep -v amdgpu > $OUT/${R}_soak_c3_300_steps_twice.txt
# kernel timelines of one REPLAYED c2 step and one of the c5 per-GPU slice (under rocprofv3 the replay's host side is slower: the
# span is longer than the bench line's step)
for C in c2 c5; do
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/tr2 -- python3 $ROOT/bench.py --config $C --no-cpu-baseline --child --steps 10 --warmup 3 > /dev/null 2>&1; cd $ROOT
python tools/trace_step.py $(ls $OUT/tr2/*/*kernel_trace.csv | head -1) -35 | cut -c1-120 > $OUT/${R}_timeline_${C}_step.txt; rm -rf $OUT/tr2
done
rm -rf $OUT/stats_* $OUT/pmc_f_* $OUT/pmc_w_* $OUT/pmc_m_* $OUT/pmc_g_*
ls -la $OUT
