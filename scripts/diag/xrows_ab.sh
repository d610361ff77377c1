# A/B of the layer-0 weight-gradient operand of a fused large step: flushed by the fused kernel (default)
# against gathered from the resident 16-bit rows by gemm_dw16_lds_kernel (V21_DW_XROWS=1).  Autoencoder stack, f16.
#   gpurun -- bash scripts/diag/xrows_ab.sh          -> gpurun_out/r5/layer0_operand_gathered_vs_flushed/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5/layer0_operand_gathered_vs_flushed
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  export V21_DW_XROWS=$f
  name=$([ $f = 1 ] && echo gathered || echo flushed)
  for n in 9216 16384 32768; do
    python3 $ROOT/scripts/train_probe.py $n f16 200 > $OUT/train_probe_b${n}_f16_$name.txt 2>&1
  done
  rocprofv3 --kernel-trace --stats -d $OUT/kt$f -o t --output-format csv -- python3 $ROOT/scripts/train_probe.py 16384 f16 100 > /dev/null 2>&1
  cp $OUT/kt$f/t_kernel_stats.csv $OUT/kernel_stats_train_b16384_f16_$name.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc${f}_$c -o p --output-format csv -- python3 $ROOT/scripts/train_probe.py 16384 f16 30 > /dev/null 2>&1
  done
  python3 $ROOT/scripts/pmc_summary.py "v21::" $OUT/pmc${f}_*/p_counter_collection.csv > $OUT/pmc_train_b16384_f16_$name.json
done
rm -rf $OUT/kt? $OUT/pmc?_*
grep -h "total" $OUT/train_probe_*.txt
