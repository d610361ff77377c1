"""Write the packaged default models of the autoencoder-based emulator
(21cmvae_amd/models/autoencoder_based_emulator/*.npz) from the weight fixture exported
from the reference's shipped Keras files (tests/golden/make_weights_fixture.py).
Data only: float32 kernels/biases + activation names."""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(ROOT, "tests", "golden", "ae_path_weights.npz"))
out = os.path.join(ROOT, "21cmvae_amd", "models", "autoencoder_based_emulator")
for stem in ("ae_emulator", "encoder", "decoder"):
    n = int(d[stem + "/n_layers"])
    blob = {"n_layers": np.array(n)}
    for i in range(n):
        blob["W%d" % i], blob["b%d" % i] = d["%s/W%d" % (stem, i)], d["%s/b%d" % (stem, i)]
        blob["act%d" % i] = np.array("linear" if i == n - 1 else "relu")
        blob["name%d" % i] = d["%s/name%d" % (stem, i)]
    np.savez_compressed(os.path.join(out, stem + ".npz"), **blob)
    print("wrote", stem)
