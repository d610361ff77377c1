#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box into gpurun_out/<tag>/ (copied to profiles/<tag>/ afterwards).
#   gpurun -- bash scripts/collect_profiles.sh r3
# kernel-trace/--stats and --pmc passes are SEPARATE runs; the profiled program comes directly after `--`.
set -u
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY=python3
# 1. headline kernel: per-kernel stats + the per-dispatch trace (100 warm-up, 100 timed, 100 with per-launch events)
rocprofv3 --kernel-trace --stats -d $OUT/bench -o b --output-format csv -- $PY $ROOT/bench.py --no-cpu-baseline --no-extras --no-train > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
cp $OUT/bench/b_kernel_stats.csv $OUT/kernel_stats_bench_f16.csv
$PY $ROOT/scripts/kstats_trace.py $OUT/bench/b_kernel_trace.csv fused_fwd 100 100 > $OUT/kernel_trace_bench_f16_timed_launches.json
# 2. HBM traffic + MFMA counters of the headline kernel: one --pmc pass per counter group
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc_$name -o p --output-format csv -- $PY $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-train > /dev/null 2>&1
done
$PY $ROOT/scripts/pmc_summary.py fused_fwd $OUT/pmc_*/p_counter_collection.csv > $OUT/pmc_fused_f16.json
# 3. training steps
rocprofv3 --kernel-trace --stats -d $OUT/t4096 -o t --output-format csv -- $PY $ROOT/scripts/train_probe.py 4096 f16 200 > $OUT/train_probe_b4096_f16.txt 2>&1
cp $OUT/t4096/t_kernel_stats.csv $OUT/kernel_stats_train_b4096_f16.csv
rocprofv3 --kernel-trace --stats -d $OUT/t256 -o t --output-format csv -- $PY $ROOT/scripts/train_probe.py 256 f32 200 > $OUT/train_probe_b256_f32.txt 2>&1
cp $OUT/t256/t_kernel_stats.csv $OUT/kernel_stats_train_b256_f32.csv
V21_CHAIN32S=0 rocprofv3 --kernel-trace --stats -d $OUT/t256r16 -o t --output-format csv -- $PY $ROOT/scripts/train_probe.py 256 f32 200 > $OUT/train_probe_b256_f32_rows16_kernel.txt 2>&1
cp $OUT/t256r16/t_kernel_stats.csv $OUT/kernel_stats_train_b256_f32_rows16_kernel.csv
V21_C32S_ROWS=8 rocprofv3 --kernel-trace --stats -d $OUT/t256r8 -o t --output-format csv -- $PY $ROOT/scripts/train_probe.py 256 f32 200 > $OUT/train_probe_b256_f32_rows8_kernel.txt 2>&1
cp $OUT/t256r8/t_kernel_stats.csv $OUT/kernel_stats_train_b256_f32_rows8_kernel.csv
rocprofv3 --kernel-trace --stats -d $OUT/t256h -o t --output-format csv -- $PY $ROOT/scripts/train_probe.py 256 f16 200 > $OUT/train_probe_b256_f16.txt 2>&1
cp $OUT/t256h/t_kernel_stats.csv $OUT/kernel_stats_train_b256_f16.csv
rocprofv3 --kernel-trace --stats -d $OUT/t16k -o t --output-format csv -- $PY $ROOT/scripts/train_probe.py 16384 f16 100 > $OUT/train_probe_b16384_f16.txt 2>&1
cp $OUT/t16k/t_kernel_stats.csv $OUT/kernel_stats_train_b16384_f16.csv
echo "train stats done"
# HBM-side bytes of every kernel of a step: one --pmc pass per counter, per workload
for wl in "4096 f16" "16384 f16" "256 f32"; do
  set -- $wl
  for c in "FETCH_SIZE" "WRITE_SIZE"; do
    rocprofv3 --pmc $c --kernel-trace -d $OUT/tpmc_$1_$2_$c -o p --output-format csv -- $PY $ROOT/scripts/train_probe.py $1 $2 30 > /dev/null 2>&1
  done
  $PY $ROOT/scripts/pmc_summary.py "v21::" $OUT/tpmc_$1_$2_*/p_counter_collection.csv > $OUT/pmc_train_b$1_$2.json
  echo "pmc $wl done"
done
# instruction mix / wait accounting / L2->CU requests of the two kernels of the f16 step at 4,096 rows
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/sq_$name -o p --output-format csv -- $PY $ROOT/scripts/train_probe.py 4096 f16 20 > /dev/null 2>&1 || echo "failed: $grp"
done
$PY $ROOT/scripts/pmc_summary.py "train_chain|dw16_adam" $OUT/sq_*/p_counter_collection.csv > $OUT/pmc_train_b4096_f16_sq_counters.json
echo "sq counters done"
# the table-driven one-launch forward (train_chain.h FORWARD mode) beside the compiled kernel, and the per-layer route
rocprofv3 --kernel-trace --stats -d $OUT/fwd -o f --output-format csv -- $PY $ROOT/scripts/forward_routes_probe.py > $OUT/forward_routes_probe.txt 2>&1
cp $OUT/fwd/f_kernel_stats.csv $OUT/kernel_stats_forward_routes.csv
# the joint step (configs[2]) against the two models one after the other
rocprofv3 --kernel-trace --stats -d $OUT/joint -o j --output-format csv -- $PY $ROOT/scripts/joint_probe.py > $OUT/joint_probe.txt 2>&1
cp $OUT/joint/j_kernel_stats.csv $OUT/kernel_stats_joint_b256_f16.csv
rocprofv3 --kernel-trace --stats -d $OUT/joint32 -o j --output-format csv -- $PY $ROOT/scripts/joint_probe.py f32 > $OUT/joint_probe_f32.txt 2>&1
cp $OUT/joint32/j_kernel_stats.csv $OUT/kernel_stats_joint_b256_f32.csv
# the sweep of 8 autoencoder configs (BASELINE configs[4]): grouped launches against the members one by one, f16 and f32
for pr in f16 f32; do
  rocprofv3 --kernel-trace --stats -d $OUT/sweep_$pr -o s --output-format csv -- $PY $ROOT/scripts/sweep_probe.py $pr 3 > $OUT/sweep_probe_$pr.txt 2>&1
  cp $OUT/sweep_$pr/s_kernel_stats.csv $OUT/kernel_stats_sweep_b256_$pr.csv
done
echo "forward + joint + sweep done"
# 3b. is the headline kernel clock-bound?  the same launches on random and on all-zero operands
cd $ROOT
$PY scripts/power_probe.py > $OUT/power_probe_fused_random_vs_zero_operands.txt 2>&1
# 4. in-kernel cycle stamps of the SHIPPED headline kernel (diagnostic build: not a benchmark)
cd $ROOT
V21_LIB=$ROOT/21cmvae_amd/libv21_stamp.so $PY scripts/diag_stamps.py f16 > $OUT/stamps_fused_f16x2sp_block_tile_cycles.txt 2>&1
# 5. per-wave stamps of the chain kernels (diagnostic build) and the two micro-benchmarks their analysis rests on
V21_LIB=$ROOT/21cmvae_amd/libv21_fine.so $PY scripts/diag/chain_wave_stamps.py 4096 f16 > $OUT/wave_stamps_chain_b4096_f16.txt 2>&1
V21_LIB=$ROOT/21cmvae_amd/libv21_fine.so $PY scripts/diag/chain_wave_stamps.py 256 f32 > $OUT/wave_stamps_chain_b256_f32.txt 2>&1
V21_C32S_ROWS=8 V21_LIB=$ROOT/21cmvae_amd/libv21_fine.so $PY scripts/diag/chain_wave_stamps.py 256 f32 > $OUT/wave_stamps_chain_b256_f32_rows8_kernel.txt 2>&1
V21_LIB=$ROOT/21cmvae_amd/libv21_fine.so $PY scripts/diag/dwadam_stamps.py 256 f32 > $OUT/phase_stamps_dwadam_b256_f32.txt 2>&1
V21_LIB=$ROOT/21cmvae_amd/libv21_fine.so V21_DW32_LDS=0 $PY scripts/diag/dwadam_stamps.py 256 f32 > $OUT/phase_stamps_dwadam_b256_f32_register_operands.txt 2>&1
V21_LIB=$ROOT/21cmvae_amd/libv21_fine.so $PY scripts/diag/dwadam_stamps.py 4096 f16 > $OUT/phase_stamps_dw16_adam_b4096_f16.txt 2>&1
[ -x scripts/diag/l1_stream_probe ] && scripts/diag/l1_stream_probe > $OUT/l1_stream_probe.txt 2>&1
[ -x scripts/diag/mfma4_rate_probe ] && scripts/diag/mfma4_rate_probe > $OUT/mfma4_rate_probe.txt 2>&1
[ -x scripts/diag/cold_stream_probe ] && scripts/diag/cold_stream_probe > $OUT/cold_stream_probe.txt 2>&1
[ -x scripts/diag/chain_loop_probe ] && scripts/diag/chain_loop_probe > $OUT/chain_loop_probe.txt 2>&1
rm -rf $OUT/t256r16 $OUT/t256r8
rm -rf $OUT/bench $OUT/pmc_*/ $OUT/t4096 $OUT/t256 $OUT/t256h $OUT/t16k $OUT/tpmc_* $OUT/sq_* $OUT/fwd $OUT/joint $OUT/joint32 $OUT/sweep_f16 $OUT/sweep_f32
ls -la $OUT
