#!/bin/bash
# Where do a kernel's wave cycles go?  Five --pmc passes (counters of one group fit one pass) over the same command.
#   bash scripts/diag/pmc_wave_anatomy.sh <out name> <kernel regex> <python script, relative to the repo> [args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; PAT=$2; SCRIPT=$ROOT/$3; shift 3
OUT=$ROOT/gpurun_out/anatomy_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o p --output-format csv -- python3 $SCRIPT "$@" > /dev/null 2>&1
done
python3 $ROOT/scripts/pmc_summary.py "$PAT" $OUT/g*/p_counter_collection.csv > $ROOT/gpurun_out/anatomy_$NAME.json
rm -rf $OUT
python3 - <<PY
import json
d = json.load(open("$ROOT/gpurun_out/anatomy_$NAME.json"))
if "_kernel" not in d: d = list(d.values())[0]
g = lambda k: d.get(k, {}).get("mean_per_dispatch", float("nan"))
wc = g("SQ_WAVE_CYCLES")
print("$NAME:", d["_kernel"])
for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_VALU_MFMA_BUSY_CYCLES"):
    print("  %-26s %14.0f  = %.3f of SQ_WAVE_CYCLES" % (k, g(k), g(k) / wc))
for k in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL", "SQ_LDS_ADDR_CONFLICT"):
    print("  %-26s %14.0f" % (k, g(k)))
PY
