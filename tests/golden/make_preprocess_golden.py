"""Golden vectors for the pre/post-processing transforms, produced by the REFERENCE.

Run once in the build container:  python tests/golden/make_preprocess_golden.py
It loads /root/reference/VeryAccurateEmulator/preprocess.py *by file path* (numpy-only
module; the package __init__, which would try a network download, is never run),
feeds it small seeded inputs and stores inputs + outputs in
tests/golden/preprocess_golden.npz.  Only data is stored - no reference source.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
synth = importlib.import_module("21cmvae_amd.synth")

spec = importlib.util.spec_from_file_location(
    "ref_preprocess", "/root/reference/VeryAccurateEmulator/preprocess.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

par_train = synth.make_params(256, seed=11, corners=True)
par_in = synth.make_params(64, seed=12)
par_in[3, 2] = 0.0  # fx == 0 branch (preprocess.py:76)
sig_train = synth.make_signals(96, seed=13)
sig_in = synth.make_signals(12, seed=14)

out = dict(
    par_train=par_train, par_in=par_in, sig_train=sig_train, sig_in=sig_in,
    par_out=ref.par_transform(par_in, par_train),
    par_out_1d=ref.par_transform(par_in[5], par_train),
    par_out_list=ref.par_transform([0.0003, 4.2, 0.0, 0.055, 1.0, 0.1, 10.0], par_train),
    par_train_out=ref.par_transform(par_train, par_train),
    pre_out=ref.preproc(sig_in, sig_train),
    unpre_out=ref.unpreproc(ref.preproc(sig_in, sig_train), sig_train),
    pre_out_f64=ref.preproc(sig_in.astype(np.float64), sig_train),
)
# float32 parameters: the reference floors and takes log10 IN THE INPUT DTYPE (preprocess.py:74-78, 89-93)
# and only then casts into its float64 result (:81-85) -- every dtype combination a caller can hand over
par_in32, par_train32 = par_in.astype(np.float32), par_train.astype(np.float32)
out.update(
    par_out_in32_tr32=ref.par_transform(par_in32, par_train32),      # float32 `parameters` and `params_train`
    par_out_in32_tr64=ref.par_transform(par_in32, par_train),        # mixed: f32 parameters, f64 training set
    par_out_in64_tr32=ref.par_transform(par_in, par_train32),        # mixed: f64 parameters, f32 training set
    par_out_1d_in32_tr32=ref.par_transform(par_in32[3], par_train32),  # 1-D float32 (the fx == 0 row)
    par_out_1d_in32_tr64=ref.par_transform(par_in32[5], par_train),
    par_train32_out=ref.par_transform(par_train32, par_train32),
)
for k in ("par_out", "par_out_1d", "pre_out", "unpre_out", "pre_out_f64", "par_out_in32_tr32", "par_out_1d_in32_tr32"):
    print(k, out[k].shape, out[k].dtype)
np.savez_compressed(os.path.join(HERE, "preprocess_golden.npz"), **out)
print("wrote preprocess_golden.npz")
