#!/bin/bash
# rocprofv3 evidence of the grouped sweep at 8 / 16 / 32 / 64 models (BASELINE configs[4] on one GPU):
# per-kernel stats and HBM-side bytes per group step -> gpurun_out/r5/ (copied to profiles/r5/).
#   gpurun -- bash scripts/profile_sweep_r5.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY=python3
for pr in f16 f32; do
  for m in 8 16 32 64; do
    rocprofv3 --kernel-trace --stats -d $OUT/sw_${pr}_m$m -o s --output-format csv -- $PY $ROOT/scripts/sweep_scaling_probe.py $pr $m > $OUT/sweep_scaling_probe_${pr}_m$m.txt 2>&1
    cp $OUT/sw_${pr}_m$m/s_kernel_stats.csv $OUT/kernel_stats_sweep_${pr}_m$m.csv
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --kernel-trace -d $OUT/swp_${pr}_m${m}_$c -o p --output-format csv -- $PY $ROOT/scripts/sweep_scaling_probe.py $pr $m > /dev/null 2>&1
    done
    $PY $ROOT/scripts/pmc_summary.py "v21::" $OUT/swp_${pr}_m${m}_*/p_counter_collection.csv > $OUT/pmc_sweep_${pr}_m$m.json
    echo "sweep $pr m$m done"
  done
done
rm -rf $OUT/sw_f16_m* $OUT/sw_f32_m* $OUT/swp_*
ls $OUT
