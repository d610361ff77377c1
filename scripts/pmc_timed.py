"""MFMA-pipe occupancy of a window of dispatches in a rocprofv3 --pmc counter CSV (the timed launches
only).  python scripts/pmc_timed.py <kernel substring> <n> <tail> <counter_collection.csv>...
bench.py --no-extras --no-train launches the headline kernel: settle, W warm-up, K timed, K with per-launch events, K with
the clock probe beside them -- the timed ones are n = K dispatches ending tail = 2 K before the last."""
import csv, json, sys, collections
pat, n, tail = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
out = {}
for path in sys.argv[4:]:
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)
    win = ids[-(n + tail):-tail] if tail else ids[-n:]
    for c in sorted({c for d in win for c in per[d]}):
        v = [per[d][c] for d in win if c in per[d]]
        out[c] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "SQ_BUSY_CYCLES" in out:
    simds = 256 * 4
    out["mfma_busy_cycles_per_simd"] = out["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_dispatch"] / simds
    out["note"] = ("SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1,024 SIMDs; SQ_BUSY_CYCLES per shader engine -- under the profiler the "
                   "dispatches are serialised and run at another clock than in the timed loop: compare cycles, not microseconds")
print(json.dumps(out, indent=1))
