"""A small Keras-2.7-style legacy .h5 model file, written with h5py (libver earliest) to
exercise 21cmvae_amd/h5lite.py without the reference's multi-MB files.
Run once:  /opt/conda/bin/python3.9 tests/golden/make_tiny_keras_h5.py
Writes tests/golden/tiny_keras_model.h5 and tiny_dataset.h5 (+ the expected arrays .npz)."""
import json, os
import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(7)
dims, names = [7, 12, 5], ["hid_0", "out"]
Ws = [rng.normal(size=(a, b)).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])]
bs = [rng.normal(size=b).astype(np.float32) for b in dims[1:]]
cfg = {"class_name": "Functional", "config": {"name": "Tiny", "layers": [
    {"class_name": "InputLayer", "config": {"batch_input_shape": [None, 7], "dtype": "float32", "name": "inp"}},
    {"class_name": "Dense", "config": {"name": "hid_0", "units": 12, "activation": "relu", "dtype": "float32"}},
    {"class_name": "Dense", "config": {"name": "out", "units": 5, "activation": "linear", "dtype": "float32"}}]}}
tc = {"loss": "mean_squared_error", "optimizer_config": {"class_name": "Adam", "config": {
    "name": "Adam", "learning_rate": 0.0005, "decay": 0.0, "beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-07, "amsgrad": False}}}
with h5py.File(os.path.join(HERE, "tiny_keras_model.h5"), "w") as f:
    f.attrs["keras_version"] = "2.7.0"
    f.attrs["backend"] = "tensorflow"
    f.attrs["model_config"] = json.dumps(cfg)
    f.attrs["training_config"] = json.dumps(tc)
    mw = f.create_group("model_weights")
    mw.attrs["layer_names"] = np.array([b"inp"] + [n.encode() for n in names])
    mw.create_group("inp").attrs["weight_names"] = np.array([], dtype="S1")
    for n, W, b in zip(names, Ws, bs):
        g = mw.create_group(n)
        g.attrs["weight_names"] = np.array([("%s/kernel:0" % n).encode(), ("%s/bias:0" % n).encode()])
        g.create_dataset("%s/kernel:0" % n, data=W)
        g.create_dataset("%s/bias:0" % n, data=b)
    ow = f.create_group("optimizer_weights")
    wn = ["Adam/iter:0"] + ["Adam/%s/%s/%s:0" % (n, p, s) for s in "mv" for n in names for p in ("kernel", "bias")]
    ow.attrs["weight_names"] = np.array([w.encode() for w in wn])
    ow.create_dataset("Adam/iter:0", data=np.int64(4242))
    ms = {}
    for s in "mv":
        for n, W, b in zip(names, Ws, bs):
            for p, a in (("kernel", W), ("bias", b)):
                v = rng.normal(size=a.shape).astype(np.float32)
                ms["%s/%s/%s" % (s, n, p)] = v
                ow.create_dataset("Adam/%s/%s/%s:0" % (n, p, s), data=v)
par = rng.uniform(1, 2, size=(10, 7))
sig = rng.normal(size=(10, 451)).astype(np.float32)
with h5py.File(os.path.join(HERE, "tiny_dataset.h5"), "w") as f:
    for k in ("train", "val", "test"):
        f.create_dataset("par_" + k, data=par + len(k))
        f.create_dataset("signal_" + k, data=sig * len(k))
np.savez(os.path.join(HERE, "tiny_h5_expected.npz"), W0=Ws[0], b0=bs[0], W1=Ws[1], b1=bs[1],
         m=np.concatenate([ms["m/%s/%s" % (n, p)].ravel() for n in names for p in ("kernel", "bias")]),
         v=np.concatenate([ms["v/%s/%s" % (n, p)].ravel() for n in names for p in ("kernel", "bias")]),
         par=par, sig=sig)
print("ok")
