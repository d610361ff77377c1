# A long fuzz campaign on seeds no test and no profile uses:  gpurun -- bash scripts/diag/fuzz_campaign.sh SEED0
set -u
S=${1:-100}
OUT=gpurun_out/fuzz_$S
mkdir -p $OUT
python scripts/diag/train_fuzz.py 300 $((S+1)) > $OUT/train.txt 2>&1; tail -n 1 $OUT/train.txt
FUZZ_BIG=1 python scripts/diag/train_fuzz.py 40 $((S+2)) > $OUT/train_big.txt 2>&1; tail -n 1 $OUT/train_big.txt
python scripts/diag/forward_fuzz.py 250 $((S+3)) > $OUT/forward.txt 2>&1; tail -n 1 $OUT/forward.txt
python scripts/diag/sweep_fuzz.py 80 $((S+4)) > $OUT/sweep.txt 2>&1; tail -n 1 $OUT/sweep.txt
python scripts/diag/joint_fuzz.py 150 $((S+5)) > $OUT/joint.txt 2>&1; tail -n 1 $OUT/joint.txt
python scripts/diag/surface_fuzz.py 60 $((S+6)) > $OUT/surface.txt 2>&1; tail -n 1 $OUT/surface.txt
python scripts/diag/dp_fuzz.py 24 $((S+7)) > $OUT/dp.txt 2>&1; tail -n 1 $OUT/dp.txt
V21_DW_XROWS=1 FUZZ_BIG=1 python scripts/diag/train_fuzz.py 40 $((S+8)) > $OUT/train_big_gathered.txt 2>&1; tail -n 1 $OUT/train_big_gathered.txt
