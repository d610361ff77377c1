# Diagnostic: SQ / TCP counters of the training chain kernel (one --pmc pass per group; gpurun -- bash scripts/diag/pmc_chain.sh)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cp_*
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVES SQ_LEVEL_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d /tmp/cp_$name -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/train_probe.py 4096 f16 20 > /dev/null 2>&1 || echo "failed: $grp"
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py "${1:-train_chain}" /tmp/cp_*/p_counter_collection.csv
