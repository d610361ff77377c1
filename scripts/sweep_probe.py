"""Sweep leg of bench.py on its own (for rocprofv3 runs)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
native = importlib.import_module("21cmvae_amd._native")
ctx = native.Context.default()
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
r = bench.sweep_leg(native, ctx, prec, epochs=int(sys.argv[2]) if len(sys.argv) > 2 else 6)
print({k: v for k, v in r.items() if k != "configs"})
