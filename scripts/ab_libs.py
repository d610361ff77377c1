import importlib, os, sys, subprocess, json
# A/B of library builds: each in its own process (V21_LIB), interleaved rounds
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(4):
    for l in libs:
        env = dict(os.environ, V21_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, "bench.py", "--no-train", "--no-extras", "--no-cpu-baseline", "--steps", "200", "--warmup", "100"], env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        res[l].append((d["ms_per_step"] * 1e3, d["roofline"]["kernel_ms_median"] * 1e3))
for l in libs:
    print(l, " | ".join("%.2f/%.2f" % t for t in res[l]))
