#!/bin/bash
# Collects the rocprofv3 evidence of round 4 on the GPU box into gpurun_out/r4/ (copied to profiles/r4/ afterwards).
#   gpurun -- bash scripts/collect_profiles_r4.sh
# kernel-trace/--stats and --pmc passes are SEPARATE runs; the profiled program comes directly after `--`.
set -u
TAG=r4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY=python3
# the build these numbers belong to: bench.py compares it with the running library
$PY $ROOT/scripts/build_id.py > $OUT/BUILD_ID
# 1. headline kernel: per-kernel stats + the per-dispatch trace (settle, 100 warm-up, 100 timed, 100 with per-launch events, 100 beside the clock probe)
rocprofv3 --kernel-trace --stats -d $OUT/bench -o b --output-format csv -- $PY $ROOT/bench.py --no-cpu-baseline --no-extras --no-train > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
cp $OUT/bench/b_kernel_stats.csv $OUT/kernel_stats_bench_f16.csv
$PY $ROOT/scripts/kstats_trace.py $OUT/bench/b_kernel_trace.csv fused_fwd 100 200 > $OUT/kernel_trace_bench_f16_timed_launches.json
echo "bench stats done"
# 2. HBM traffic + MFMA counters of the headline kernel: one --pmc pass per counter group; MFMA-pipe occupancy of the TIMED launches
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc_$name -o p --output-format csv -- $PY $ROOT/bench.py --steps 20 --warmup 10 --settle 0.05 --no-cpu-baseline --no-extras --no-train > /dev/null 2>&1
done
$PY $ROOT/scripts/pmc_summary.py fused_fwd $OUT/pmc_*/p_counter_collection.csv > $OUT/pmc_fused_f16.json
$PY $ROOT/scripts/pmc_timed.py fused_fwd 20 40 $OUT/pmc_SQ_INSTS_MFMA*/p_counter_collection.csv $OUT/pmc_GRBM*/p_counter_collection.csv > $OUT/pmc_fused_f16_timed_launches_mfma_busy.json
echo "bench pmc done"
# 3. training steps: the two launches of a step (+ the large-step route of r4: pack, fused_train, split-K gradients, Adam)
for wl in "4096 f16 200" "256 f32 200" "256 f16 200" "16384 f16 100" "32768 f16 60"; do
  set -- $wl
  rocprofv3 --kernel-trace --stats -d $OUT/t$1$2 -o t --output-format csv -- $PY $ROOT/scripts/train_probe.py $1 $2 $3 > $OUT/train_probe_b$1_$2.txt 2>&1
  cp $OUT/t$1$2/t_kernel_stats.csv $OUT/kernel_stats_train_b$1_$2.csv
done
V21_FUSED_TRAIN=0 rocprofv3 --kernel-trace --stats -d $OUT/t32kc -o t --output-format csv -- $PY $ROOT/scripts/train_probe.py 32768 f16 60 > $OUT/train_probe_b32768_f16_chain_route.txt 2>&1
cp $OUT/t32kc/t_kernel_stats.csv $OUT/kernel_stats_train_b32768_f16_chain_route.csv
echo "train stats done"
for wl in "4096 f16" "16384 f16" "32768 f16" "256 f32"; do
  set -- $wl
  for c in "FETCH_SIZE" "WRITE_SIZE"; do
    rocprofv3 --pmc $c --kernel-trace -d $OUT/tpmc_$1_$2_$c -o p --output-format csv -- $PY $ROOT/scripts/train_probe.py $1 $2 30 > /dev/null 2>&1
  done
  $PY $ROOT/scripts/pmc_summary.py "v21::" $OUT/tpmc_$1_$2_*/p_counter_collection.csv > $OUT/pmc_train_b$1_$2.json
  echo "pmc $wl done"
done
# 4. forward routes (compiled, run-time instantiated, table-driven, per-layer), joint step, sweeps
rocprofv3 --kernel-trace --stats -d $OUT/fwd -o f --output-format csv -- $PY $ROOT/scripts/forward_routes_probe.py > $OUT/forward_routes_probe.txt 2>&1
cp $OUT/fwd/f_kernel_stats.csv $OUT/kernel_stats_forward_routes.csv
rocprofv3 --kernel-trace --stats -d $OUT/joint -o j --output-format csv -- $PY $ROOT/scripts/joint_probe.py > $OUT/joint_probe.txt 2>&1
cp $OUT/joint/j_kernel_stats.csv $OUT/kernel_stats_joint_b256_f16.csv
rocprofv3 --kernel-trace --stats -d $OUT/joint32 -o j --output-format csv -- $PY $ROOT/scripts/joint_probe.py f32 > $OUT/joint_probe_f32.txt 2>&1
cp $OUT/joint32/j_kernel_stats.csv $OUT/kernel_stats_joint_b256_f32.csv
for pr in f16 f32; do
  rocprofv3 --kernel-trace --stats -d $OUT/sweep_$pr -o s --output-format csv -- $PY $ROOT/scripts/sweep_probe.py $pr 3 > $OUT/sweep_probe_$pr.txt 2>&1
  cp $OUT/sweep_$pr/s_kernel_stats.csv $OUT/kernel_stats_sweep_b256_$pr.csv
done
echo "forward + joint + sweep done"
# 5. r4 diagnostics: the headline kernel with half the LDS bytes per MFMA + the clock beside both forms; the run-time kernels
# against the oracle; the fused training kernel against the chain route
cd $ROOT
$PY scripts/diag/k1_wide_probe.py > $OUT/k1_wide_variant_and_clock_probe.txt 2>&1
$PY scripts/diag/jit_parity_probe.py > $OUT/jit_parity_probe.txt 2>&1
$PY scripts/diag/jit_fuzz.py 40 1 > $OUT/jit_fuzz_40_random_stacks.txt 2>&1
# (the library's choice of fused training kernel: 16 rows per wave below 24,576 rows per step, 32 above -- and the other one of each, forced)
for n in 9216 12288 16384 32768; do $PY scripts/diag/fused_train_probe.py $n f16 > $OUT/fused_train_probe_b${n}_f16.txt 2>&1; done
V21_FUSED_TRAIN16=0 $PY scripts/diag/fused_train_probe.py 16384 f16 > $OUT/fused_train_probe_b16384_f16_forced_32_rows_per_wave.txt 2>&1
V21_FUSED_TRAIN16=1 $PY scripts/diag/fused_train_probe.py 32768 f16 > $OUT/fused_train_probe_b32768_f16_forced_16_rows_per_wave.txt 2>&1
$PY scripts/power_probe.py > $OUT/power_probe_fused_random_vs_zero_operands.txt 2>&1
# 5b. the round's fuzzers (random cases against the float64 oracle and a bitwise twin; every line ends in OK / BAD / refused)
$PY scripts/diag/train_fuzz.py 150 11 > $OUT/train_fuzz_150_cases.txt 2>&1
FUZZ_BIG=1 $PY scripts/diag/train_fuzz.py 20 12 > $OUT/train_fuzz_large_steps_20_cases.txt 2>&1
$PY scripts/diag/forward_fuzz.py 100 13 > $OUT/forward_fuzz_100_cases.txt 2>&1
$PY scripts/diag/sweep_fuzz.py 60 14 > $OUT/sweep_fuzz_60_cases.txt 2>&1
$PY scripts/diag/joint_fuzz.py 80 15 > $OUT/joint_fuzz_80_cases.txt 2>&1
$PY scripts/diag/surface_fuzz.py 40 16 > $OUT/class_surface_fuzz_40_cases.txt 2>&1
$PY scripts/diag/dp_fuzz.py 36 2 > $OUT/dp_fuzz_36_cases_2_to_4_ranks_on_one_gpu.txt 2>&1
echo "fuzzers done"
# 6. the widest hidden layer alone (7 -> 352 x 6 -> 9): duration by rocprofv3, MFMA-pipe counters of its 200 timed launches
cd /tmp
for pr in f16 bf16; do
  $PY $ROOT/scripts/hidden_layer_probe.py $pr > $OUT/hidden_layers_352_${pr}_probe.txt 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/hid_kt_$pr -o k --output-format csv -- $PY $ROOT/scripts/hidden_layer_probe.py $pr > /dev/null 2>&1
  cp $OUT/hid_kt_$pr/k_kernel_stats.csv $OUT/kernel_stats_hidden_layers_352_$pr.csv
  for grp in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
    n=$(echo $grp | cut -d" " -f1)
    rocprofv3 --pmc $grp --kernel-trace -d $OUT/hid_pmc_${pr}_$n -o p --output-format csv -- $PY $ROOT/scripts/hidden_layer_probe.py $pr > /dev/null 2>&1
  done
  $PY $ROOT/scripts/pmc_timed.py fused_fwd 200 0 $OUT/hid_pmc_${pr}_SQ_INSTS_MFMA/p_counter_collection.csv $OUT/hid_pmc_${pr}_GRBM_GUI_ACTIVE/p_counter_collection.csv > $OUT/hidden_layers_352_${pr}_mfma_busy.json
done
cd $ROOT
echo "hidden layers done"
# 7. where a workgroup of the fused training kernel spends its cycles (diagnostic build `make -C 21cmvae_amd/csrc tstamp`, if present), and
# the one-wave-per-SIMD microbenchmarks DESIGN.md section 3 K3-fused quotes (compiled here: hipcc is on the box)
if [ -f $ROOT/21cmvae_amd/libv21_tstamp.so ]; then
  for k in 0 1; do for n in 16384 32768; do
    V21_FUSED_TRAIN16=$k V21_LIB=$ROOT/21cmvae_amd/libv21_tstamp.so $PY scripts/diag/fused_train_stamps.py $n > $OUT/fused_train_stamps_b${n}_f16_$((32 - 16 * k))_rows_per_wave.txt 2>&1
  done; done
fi
for pb in mfma_chain_probe lds_read_probe mfma_issue_probe; do
  hipcc -std=c++20 --offload-arch=gfx950 -O3 -o /tmp/$pb scripts/diag/$pb.hip > /dev/null 2>&1 && timeout -k 5 120 /tmp/$pb > $OUT/microbench_$pb.txt 2>&1
done
echo "stamps + microbenchmarks done"
rm -rf $OUT/bench $OUT/hid_kt_* $OUT/hid_pmc_* $OUT/pmc_*/ $OUT/t*f16 $OUT/t*f32 $OUT/t32kc $OUT/tpmc_* $OUT/fwd $OUT/joint $OUT/joint32 $OUT/sweep_f16 $OUT/sweep_f32
ls -la $OUT
