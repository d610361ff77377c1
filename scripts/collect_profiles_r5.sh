#!/bin/bash
# Collects the rocprofv3 evidence of round 5 on the GPU box into gpurun_out/r5/ (copied to profiles/r5/ afterwards).
#   gpurun -- bash scripts/collect_profiles_r5.sh [part]      part: all (default) | bench | train | rest | fuzz
# kernel-trace/--stats and --pmc passes are SEPARATE runs; the profiled program comes directly after `--`.
set -u
TAG=r5
PART=${1:-all}
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY=python3
# the build these numbers belong to: bench.py compares it with the running library
$PY $ROOT/scripts/build_id.py > $OUT/BUILD_ID
if want bench; then
# 1. headline kernel: per-kernel stats + the per-dispatch trace (settle, 100 warm-up, 100 timed, 100 with per-launch events; then the
# clock-stamped instantiation -- another kernel name -- settle, 100 warm-up, 100 stamped)
rocprofv3 --kernel-trace --stats -d $OUT/bench -o b --output-format csv -- $PY $ROOT/bench.py --no-cpu-baseline --no-extras --no-train > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
cp $OUT/bench/b_kernel_stats.csv $OUT/kernel_stats_bench_f16.csv
$PY $ROOT/scripts/kstats_trace.py $OUT/bench/b_kernel_trace.csv "x2sp>" 100 100 > $OUT/kernel_trace_bench_f16_timed_launches.json
echo "bench stats done"
# 2. HBM traffic + MFMA counters of the headline kernel: one --pmc pass per counter group; MFMA-pipe occupancy of the TIMED launches
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc_$name -o p --output-format csv -- $PY $ROOT/bench.py --steps 20 --warmup 10 --settle 0.05 --no-cpu-baseline --no-extras --no-train > /dev/null 2>&1
done
$PY $ROOT/scripts/pmc_summary.py "x2sp>" $OUT/pmc_*/p_counter_collection.csv > $OUT/pmc_fused_f16.json
$PY $ROOT/scripts/pmc_timed.py "x2sp>" 20 20 $OUT/pmc_SQ_INSTS_MFMA*/p_counter_collection.csv $OUT/pmc_GRBM*/p_counter_collection.csv > $OUT/pmc_fused_f16_timed_launches_mfma_busy.json
echo "bench pmc done"
fi
if want train; then
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
# r5: where the 4,096-row f16 step's cycles go (the per-layer budget the item is closed with)
$PY $ROOT/scripts/diag/step_budget.py 4096 f16 > $OUT/cycle_budget_train_b4096_f16.txt 2>&1
$PY $ROOT/scripts/diag/step_budget.py 256 f16 > $OUT/cycle_budget_train_b256_f16.txt 2>&1
fi
if want rest; then
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
# r5: configs[4] on one GPU -- the grouped sweep at 8 / 16 / 32 / 64 models: per-kernel stats + HBM-side bytes per group step
bash $ROOT/scripts/profile_sweep_r5.sh > $OUT/profile_sweep.log 2>&1
cd /tmp
# r5: the shader clock from the timed kernel itself, against the time since the first launch
$PY $ROOT/scripts/diag/clock_drift_probe.py 3 f16 > $OUT/clock_in_kernel_vs_time_f16.txt 2>&1
echo "forward + joint + sweep done"
# 5. diagnostics: the run-time kernels against the oracle; the fused training kernel against the chain route
cd $ROOT
$PY scripts/diag/jit_parity_probe.py > $OUT/jit_parity_probe.txt 2>&1
$PY scripts/diag/jit_fuzz.py 40 1 > $OUT/jit_fuzz_40_random_stacks.txt 2>&1
# (the library's choice of fused training kernel: 16 rows per wave below 24,576 rows per step, 32 above -- and the other one of each, forced)
for n in 9216 12288 16384 32768; do $PY scripts/diag/fused_train_probe.py $n f16 > $OUT/fused_train_probe_b${n}_f16.txt 2>&1; done
V21_FUSED_TRAIN16=0 $PY scripts/diag/fused_train_probe.py 16384 f16 > $OUT/fused_train_probe_b16384_f16_forced_32_rows_per_wave.txt 2>&1
V21_FUSED_TRAIN16=1 $PY scripts/diag/fused_train_probe.py 32768 f16 > $OUT/fused_train_probe_b32768_f16_forced_16_rows_per_wave.txt 2>&1
# r5, VERDICT r4 item 7: the layer-0 gradient operand flushed (default) against gathered from the resident rows (V21_DW_XROWS=1)
bash $ROOT/scripts/diag/xrows_ab.sh > $OUT/layer0_operand_ab.log 2>&1
cd $ROOT
$PY scripts/power_probe.py > $OUT/power_probe_fused_random_vs_zero_operands.txt 2>&1
fi
if want fuzz; then
cd $ROOT
# 5b. the round's fuzzers (random cases against the float64 oracle and a bitwise twin; every line ends in OK / BAD / refused)
# (other seeds than the slices tests/test_fuzz_gpu.py runs under pytest -m gpu)
$PY scripts/diag/train_fuzz.py 120 31 > $OUT/train_fuzz_120_cases.txt 2>&1
FUZZ_BIG=1 $PY scripts/diag/train_fuzz.py 16 32 > $OUT/train_fuzz_large_steps_16_cases.txt 2>&1
$PY scripts/diag/forward_fuzz.py 80 33 > $OUT/forward_fuzz_80_cases.txt 2>&1
$PY scripts/diag/sweep_fuzz.py 40 34 > $OUT/sweep_fuzz_40_cases_up_to_64_members.txt 2>&1
$PY scripts/diag/joint_fuzz.py 50 35 > $OUT/joint_fuzz_50_cases.txt 2>&1
$PY scripts/diag/surface_fuzz.py 24 36 > $OUT/class_surface_fuzz_24_cases.txt 2>&1
$PY scripts/diag/dp_fuzz.py 12 37 > $OUT/dp_fuzz_12_cases_2_to_4_ranks_on_one_gpu.txt 2>&1
# the f32 joint step against a separate trainer on the oracle's latents: the relative loss difference by steps per epoch
FUZZ_ONLY=115 $PY scripts/diag/joint_case115_r4.py 200 105 > $OUT/joint_f32_drift_by_step_count.txt 2>&1
echo "fuzzers done"
fi
if want rest; then
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
fi
rm -rf $OUT/bench $OUT/hid_kt_* $OUT/hid_pmc_* $OUT/pmc_*/ $OUT/t[0-9]*f16 $OUT/t[0-9]*f32 $OUT/t32kc $OUT/tpmc_* $OUT/fwd $OUT/joint $OUT/joint32 $OUT/sweep_f16 $OUT/sweep_f32
ls -la $OUT
