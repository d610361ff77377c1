"""BASELINE configs[4] on one GPU (r5): the grouped sweep at 8 / 16 / 32 / 64 models in ONE group -- bench.py's
sweep_scaling leg on its own (also what scripts/collect_profiles_r5.sh puts under rocprofv3).
  python scripts/sweep_scaling_probe.py [f16|bf16|f32] [counts, e.g. 8,16,32,64]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
native = importlib.import_module("21cmvae_amd._native")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
counts = tuple(int(c) for c in sys.argv[2].split(",")) if len(sys.argv) > 2 else (8, 16, 32, 64)
ctx = native.Context.default()
out = bench.sweep_scaling_leg(native, ctx, prec, counts=counts)
for g in out["groups"]:
    rf = g["roofline"]
    print("%s models %2d: %8.0f model-steps/s  %7.1f us per group step  %.2fx the first group's rate  algorithmic %.0f GB/s = %.3f of 8 TB/s (MFMA: %.4f of peak)  params %d"
          % (prec, g["models"], g["model_steps_per_s"], g["ms_per_group_step"] * 1e3, g["vs_8_models"], rf["achieved"], rf["frac"], rf["mfma_frac"], g["parameters_of_the_group"]))
print(json.dumps(out))
