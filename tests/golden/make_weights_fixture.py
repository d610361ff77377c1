"""Export the reference's shipped Keras weight files to a plain .npz fixture.

Run ONCE in the build container (needs h5py, which only the conda interpreter
has):  /opt/conda/bin/python3.9 tests/golden/make_weights_fixture.py

Inputs (data files shipped by the reference, read-only):
  /root/reference/VeryAccurateEmulator/models/autoencoder_based_emulator/
      {ae_emulator,encoder,decoder,autoencoder}.h5
Output: tests/golden/ae_path_weights.npz  (float32 kernels (in,out) + biases in
layer order, plus the optimizer `iter` / learning-rate scalars of the two
files that carry a training_config).  Nothing is executed from the files:
h5py only reads datasets and attributes.
"""
import json
import os
import sys

import h5py
import numpy as np

SRC = "/root/reference/VeryAccurateEmulator/models/autoencoder_based_emulator/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ae_path_weights.npz")


def read_model(path):
    out = {}
    with h5py.File(path, "r") as f:
        mw = f["model_weights"]
        names = [n.decode() if isinstance(n, bytes) else n for n in mw.attrs["layer_names"]]
        li = 0
        for name in names:
            g = mw[name]
            wn = [w.decode() if isinstance(w, bytes) else w for w in g.attrs["weight_names"]]
            if not wn:
                continue
            kern = np.asarray(g[wn[0]], dtype=np.float32)
            bias = np.asarray(g[wn[1]], dtype=np.float32)
            out["W%d" % li] = kern
            out["b%d" % li] = bias
            out["name%d" % li] = np.array(name)
            li += 1
        out["n_layers"] = np.array(li)
        if "optimizer_weights" in f:
            out["adam_iter"] = np.asarray(f["optimizer_weights/Adam/iter:0"])
            tc = f.attrs["training_config"]
            tc = json.loads(tc.decode() if isinstance(tc, bytes) else tc)
            cfg = tc["optimizer_config"]["config"]
            for k in ("learning_rate", "beta_1", "beta_2", "epsilon"):
                out["adam_" + k] = np.array(cfg[k], dtype=np.float64)
            out["loss_name"] = np.array(tc["loss"])
    return out


def main():
    blob = {}
    for stem in ("ae_emulator", "encoder", "decoder", "autoencoder"):
        for k, v in read_model(SRC + stem + ".h5").items():
            blob[stem + "/" + k] = v
    # autoencoder.h5 repeats encoder.h5 + decoder.h5: keep the fixture small by
    # checking bit-identity here and storing only its scalars.
    chain = [("encoder", 0), ("encoder", 1), ("decoder", 0), ("decoder", 1), ("decoder", 2)]
    for i, (stem, j) in enumerate(chain):
        for p in "Wb":
            a = blob.pop("autoencoder/%s%d" % (p, i))
            assert np.array_equal(a, blob["%s/%s%d" % (stem, p, j)]), (stem, p, i)
    blob["autoencoder/equals_encoder_plus_decoder"] = np.array(True)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(blob), "arrays")


if __name__ == "__main__":
    sys.exit(main())
