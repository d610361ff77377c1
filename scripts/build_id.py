"""A hash of every source the HIP library is built from (csrc/*.h, *.hip, *.cpp, Makefile, include/*.h): profiles/<tag>/BUILD_ID
says which build a profile belongs to, bench.py says whether the running tree still is that build."""
import glob, hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_id():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "21cmvae_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "21cmvae_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "21cmvae_amd", "csrc", "*.cpp")) + [os.path.join(ROOT, "21cmvae_amd", "csrc", "Makefile")] +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(build_id())
