"""CPU: the pure-Python HDF5 / Keras-legacy-H5 reader."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, pkg

h5 = pkg("h5lite")


def test_keras_model_file_roundtrip():
    exp = np.load(os.path.join(GOLDEN, "tiny_h5_expected.npz"))
    info = h5.read_keras_h5(os.path.join(GOLDEN, "tiny_keras_model.h5"))
    assert [(l[0], l[3]) for l in info["layers"]] == [("hid_0", "relu"), ("out", "linear")]
    np.testing.assert_array_equal(info["layers"][0][1], exp["W0"])
    np.testing.assert_array_equal(info["layers"][0][2], exp["b0"])
    np.testing.assert_array_equal(info["layers"][1][1], exp["W1"])
    assert info["optimizer"]["iter"] == 4242
    np.testing.assert_array_equal(info["optimizer"]["m"], exp["m"])  # Keras order: all m, then all v
    np.testing.assert_array_equal(info["optimizer"]["v"], exp["v"])
    assert info["optimizer"]["config"]["learning_rate"] == 0.0005
    assert info["config"]["config"]["name"] == "Tiny"


def test_dataset_file_like_h5py():
    exp = np.load(os.path.join(GOLDEN, "tiny_h5_expected.npz"))
    with h5.File(os.path.join(GOLDEN, "tiny_dataset.h5"), "r") as hf:
        assert sorted(hf.keys()) == ["par_test", "par_train", "par_val", "signal_test", "signal_train", "signal_val"]
        pt = hf["par_train"][:]
        st = hf["signal_test"][:]
        assert pt.dtype == np.float64 and st.dtype == np.float32
        np.testing.assert_array_equal(pt, exp["par"] + 5)
        np.testing.assert_array_equal(st, exp["sig"] * 4)
        np.testing.assert_array_equal(hf["signal_val"][2:4], (exp["sig"] * 3)[2:4])


def test_rejects_non_hdf5_and_missing(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all")
    with pytest.raises(IOError):
        h5.File(str(p))
    with pytest.raises(IOError):
        h5.read_keras_h5(os.path.join(GOLDEN, "tiny_dataset.h5"))  # no model_weights group


@pytest.mark.skipif(not os.path.exists("/root/reference/VeryAccurateEmulator/models"), reason="reference not mounted")
def test_reads_the_reference_shipped_files(shipped):
    base = "/root/reference/VeryAccurateEmulator/models/autoencoder_based_emulator/"
    for stem in ("ae_emulator", "encoder", "decoder"):
        info = h5.read_keras_h5(base + stem + ".h5")
        Ws, bs = shipped[stem]
        assert len(info["layers"]) == len(Ws)
        for (name, k, b, a), W, bb in zip(info["layers"], Ws, bs):
            np.testing.assert_array_equal(k, W)
            np.testing.assert_array_equal(b, bb)
    assert h5.read_keras_h5(base + "ae_emulator.h5")["optimizer"]["iter"] == 17568
