# Diagnostic: SQ counters of the headline kernel (one --pmc pass per group; gpurun -- bash scripts/diag/pmc_fused.sh)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cf_*
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d /tmp/cf_$name -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --settle 0.02 --no-cpu-baseline --no-extras --no-train > /dev/null 2>&1 || echo "failed: $grp"
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py "fused_fwd" /tmp/cf_*/p_counter_collection.csv
