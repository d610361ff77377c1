"""Duration statistics of a window of n dispatches of a kernel in a rocprofv3 kernel-trace CSV, `tail` dispatches
before the end (bench.py launches W warm-up, K timed, then K more with an event pair each: the timed ones are
n = K, tail = K):  python scripts/kstats_trace.py <kernel_trace.csv> <name substring> <n> [tail]"""
import csv, json, sys
import numpy as np
path, pat, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
tail = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in (rows[-(n + tail):-tail] if tail else rows[-n:])], dtype=np.float64) / 1e3
name = rows[-1]["Kernel_Name"] if rows else "?"
print(json.dumps({"kernel": name[:100], "dispatches_in_trace": len(rows), "window": "%d dispatches ending %d before the last" % (len(d), tail),
                  "us_mean": float(d.mean()), "us_median": float(np.median(d)), "us_min": float(d.min()), "us_max": float(d.max()),
                  "note": "kernel durations (begin to end of each dispatch, no launch gaps) of bench.py's timed launches, from rocprofv3 --kernel-trace"}, indent=1))
