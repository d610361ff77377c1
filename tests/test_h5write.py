"""The Keras legacy-H5 writer (21cmvae_amd/h5write.py) against the reader (h5lite) -- CPU only -- and, when
an interpreter with h5py exists on this machine (the build container has one under /opt/conda), against the
real HDF5 library.  The reference loads such files with tf.keras.models.load_model (emulator.py:335-337)."""
import importlib
import os
import subprocess

import numpy as np
import pytest

hw = importlib.import_module("21cmvae_amd.h5write")
hl = importlib.import_module("21cmvae_amd.h5lite")

H5PY_PYTHON = "/opt/conda/bin/python3.9"


def _toy(seed=0):
    rng = np.random.default_rng(seed)
    names = ["em_hidden_layer_0", "em_hidden_layer_1", "dense"]
    Ws = [rng.normal(size=s).astype(np.float32) for s in ((7, 16), (16, 8), (8, 451))]
    bs = [rng.normal(size=w.shape[1]).astype(np.float32) for w in Ws]
    P = sum(w.size + b.size for w, b in zip(Ws, bs))
    opt = {"iter": 17568, "m": rng.normal(size=P).astype(np.float32), "v": rng.uniform(size=P).astype(np.float32),
           "config": {"learning_rate": 2.7813e-4, "beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-7}}
    return names, Ws, bs, opt


def test_round_trip_through_the_reader(tmp_path):
    names, Ws, bs, opt = _toy()
    p = str(tmp_path / "model.h5")
    hw.write_keras_h5(p, names, Ws, bs, ["relu", "relu", "linear"], "emulator", opt)
    info = hl.read_keras_h5(p)
    assert [l[0] for l in info["layers"]] == names
    assert [l[3] for l in info["layers"]] == ["relu", "relu", "linear"]
    for (n, k, b, a), W, B in zip(info["layers"], Ws, bs):
        np.testing.assert_array_equal(k, W); np.testing.assert_array_equal(b, B)
    o = info["optimizer"]
    assert o["iter"] == 17568 and abs(o["config"]["learning_rate"] - 2.7813e-4) < 1e-12
    np.testing.assert_array_equal(o["m"], opt["m"]); np.testing.assert_array_equal(o["v"], opt["v"])
    with hl.File(p) as f:  # the scalar iteration counter keeps Keras' shape ()
        assert f["optimizer_weights"]["Adam/iter:0"].shape in ((), (1,))
        assert hl._s(f.attrs["keras_version"]) == "2.7.0"


def test_generic_tree_round_trip(tmp_path):
    """Nested groups, many links in one group, float64/int32 data, numeric and string-list attributes."""
    f = hw.FileW()
    f.attrs["title"] = "t"
    g = f.create_group("a/b")
    g.attrs["names"] = ["x", "longer_name", ""]
    g.attrs["scale"] = np.float32(2.5)
    data = {}
    for i in range(40):
        data["d%02d" % i] = np.arange(i + 1, dtype=np.float64) * 0.5
        g.create_dataset("d%02d" % i, data["d%02d" % i])
    f.create_dataset("ints", np.arange(6, dtype=np.int32).reshape(2, 3))
    p = str(tmp_path / "tree.h5")
    f.write(p)
    with hl.File(p) as r:
        np.testing.assert_array_equal(r["ints"][:], np.arange(6).reshape(2, 3))
        gb = r["a"]["b"]
        assert sorted(gb.keys()) == sorted(data)
        for k, v in data.items():
            np.testing.assert_array_equal(gb[k][:], v)
        assert [hl._s(n) for n in np.atleast_1d(gb.attrs["names"])] == ["x", "longer_name", ""]
        assert float(np.asarray(gb.attrs["scale"]).reshape(-1)[0]) == 2.5
    with pytest.raises(ValueError):
        big = hw.FileW()
        for i in range(70):
            big.create_dataset("d%d" % i, np.zeros(1, np.float32))
        big.write(str(tmp_path / "big.h5"))


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py on this machine")
def test_real_hdf5_library_reads_the_file(tmp_path):
    names, Ws, bs, opt = _toy(3)
    p = str(tmp_path / "model.h5")
    hw.write_keras_h5(p, names, Ws, bs, ["relu", "relu", "linear"], "emulator", opt)
    code = (
        "import h5py, json, numpy as np, sys\n"
        "f = h5py.File(sys.argv[1], 'r')\n"
        "mc = f.attrs['model_config']; mc = mc.decode() if isinstance(mc, bytes) else mc\n"
        "cfg = json.loads(mc)\n"
        "out = {'layers': [l['config'].get('activation') for l in cfg['config']['layers'] if l['class_name'] == 'Dense']}\n"
        "mw = f['model_weights']\n"
        "out['names'] = [n.decode() for n in mw.attrs['layer_names']]\n"
        "out['sums'] = [float(np.abs(mw[n][n]['kernel:0'][:]).sum()) for n in out['names']]\n"
        "out['shapes'] = [list(mw[n][n]['bias:0'].shape) for n in out['names']]\n"
        "ow = f['optimizer_weights']\n"
        "out['iter'] = int(np.asarray(ow['Adam/iter:0'][()]).reshape(-1)[0]); out['iter_shape'] = list(ow['Adam/iter:0'].shape)\n"
        "out['vsum'] = float(ow['Adam/dense/bias/v:0'][:].sum())\n"
        "print(json.dumps(out))\n")
    try:
        r = subprocess.run([H5PY_PYTHON, "-c", code, p], capture_output=True, text=True, timeout=120)
    except OSError as e:  # pragma: no cover
        pytest.skip("cannot run %s: %s" % (H5PY_PYTHON, e))
    if r.returncode != 0 and "No module named" in r.stderr:
        pytest.skip("h5py not importable there")
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["names"] == names and out["layers"] == ["relu", "relu", "linear"]
    np.testing.assert_allclose(out["sums"], [float(np.abs(w).sum()) for w in Ws], rtol=1e-6)
    assert out["shapes"] == [[16], [8], [451]]
    assert out["iter"] == 17568 and out["iter_shape"] == []
    np.testing.assert_allclose(out["vsum"], float(opt["v"][-451:].sum()), rtol=1e-6)


def test_dataset_file_round_trip(tmp_path, monkeypatch):
    """synth.save_dataset -> the file the no-argument constructors look for ($V21_DATASET): the six arrays come
    back bit-identical through emulator.load_dataset (the reference reads them at import, emulator.py:198-204)."""
    synth = importlib.import_module("21cmvae_amd.synth")
    emu = importlib.import_module("21cmvae_amd.emulator")
    monkeypatch.setattr(emu, "_dataset", emu._dataset)  # load_dataset caches: undo it at teardown
    data = synth.make_dataset(60, 20, 10, seed=4)
    p = synth.save_dataset(str(tmp_path / "dataset_21cmVAE.h5"), data)
    got = emu.load_dataset(p)
    assert sorted(got) == sorted(data)
    for k in data:
        assert got[k].dtype == data[k].dtype and got[k].shape == data[k].shape
        np.testing.assert_array_equal(got[k], data[k])
    # the shared arrays are read-only, so the statistics cache needs no per-call checksum of them (preprocess._cached);
    # with the flag flipped back the buffer is hashed again and an in-place edit is noticed
    pp = importlib.import_module("21cmvae_amd.preprocess")
    sig = got["signal_train"]
    assert pp._fingerprint(sig)[2] == "frozen" and pp._fingerprint(got["par_train"])[2] == "frozen"
    with pytest.raises(ValueError):
        sig[0, 0] = 1.0
    s0 = pp.SignalStats.of(sig)
    assert pp.SignalStats.of(sig) is s0
    sig.flags.writeable = True
    assert pp._fingerprint(sig)[2] != "frozen"
    sig[0, 0] += 1.0
    s1 = pp.SignalStats.of(sig)
    assert s1 is not s0 and s1.mean[0] != s0.mean[0]
