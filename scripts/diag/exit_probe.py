import importlib, sys, numpy as np
sys.path.insert(0, "/root/repo")
native = importlib.import_module("21cmvae_amd._native")
ctx = native.Context(0)
st = native.Stack(ctx, [7, 64, 128, 451], [1, 1, 0])
st.set_weights(np.zeros(st.num_params, np.float32))
which = sys.argv[1]
if which == "torch_after":
    import torch
    print("torch imported after", torch.cuda.is_available())
elif which == "train_jit":
    tr = native.Trainer(st, "f16", 16384)   # asks for a run-time kernel at creation (prebuilt for this stack: ready at once)
    st2 = native.Stack(ctx, [7, 96, 200, 451], [1, 1, 0]); st2.set_weights(np.zeros(st2.num_params, np.float32))
    tr2 = native.Trainer(st2, "f16", 16384)  # not prebuilt: a compile starts in the background; exit while it runs
print("done", which)
