"""CPU: the run-time instantiation of the fused forward kernel (csrc/jit.hip) up to the point where a GPU is needed --
hiprtc compiles for gfx950 without one.  What the reference side of this is: `_gen_model` accepts ANY hidden_dims
(/root/reference/VeryAccurateEmulator/emulator.py:12-48), so every stack must get the fast kernel, not only the four
compiled into the library (csrc/archs.h)."""
import ctypes.util
import os
import subprocess

import pytest

from conftest import ROOT, pkg

HAVE_HIPRTC = os.path.exists("/opt/rocm/lib/libhiprtc.so") or ctypes.util.find_library("hiprtc") is not None
needs_hiprtc = pytest.mark.skipif(not HAVE_HIPRTC, reason="libhiprtc is not installed here")


def _prebuild_in_a_fresh_process(dims, act, prec, d):
    """hiprtc's LLVM must be the only one in its process (pytest has torch's loaded by now: its option table lacks the AMDGPU
    flags and LLVM exits): a fresh interpreter, as __graft_entry__.build() uses -- the library itself compiles in `v21_jitc`."""
    import sys
    code = ("import importlib, sys; sys.path.insert(0, %r); n = importlib.import_module('21cmvae_amd._native'); "
            "n.jit_prebuild(%r, %r, %r, %r)" % (ROOT, dims, act, prec, d))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]


@needs_hiprtc
def test_prebuild_writes_a_code_object_and_finds_it_again(tmp_path):
    d = str(tmp_path / "kc")
    _prebuild_in_a_fresh_process([7, 48, 451], [1, 0], "f16", d)
    files = os.listdir(d)
    assert len(files) == 1 and files[0].startswith("fused_7x48x451_a10_PrecF16x2sp_") and files[0].endswith(".v21k")
    blob = open(os.path.join(d, files[0]), "rb").read()
    assert blob[:8] == b"V21KOBJ1" and b"fused_fwd" in blob[:200] and b"\x7fELF" in blob   # header, mangled kernel name, code object
    mtime = os.path.getmtime(os.path.join(d, files[0]))
    _prebuild_in_a_fresh_process([7, 48, 451], [1, 0], "f16", d)                            # same sources, same options: found, not rebuilt
    assert os.path.getmtime(os.path.join(d, files[0])) == mtime
    _prebuild_in_a_fresh_process([7, 48, 451], [1, 0], "f32", d)                            # another precision: another kernel
    assert len(os.listdir(d)) == 2


def test_stacks_the_fused_kernel_cannot_express_are_refused(tmp_path):
    native = pkg("_native")
    for dims, act, why in (([7, 32, 9], [1, 1], "output layer is linear"),                  # ReLU on the output layer
                           ([451, 64, 9, 32, 451], [1, 2, 1, 0], "variational"),            # (z_mean | z_log_var) head
                           ([7, 16, 451], [1, 0], "too narrow")):                           # 16 features into a multi-tile output layer
        with pytest.raises(native.EngineError, match=why):
            native.jit_prebuild(dims, act, "f16", str(tmp_path))


@needs_hiprtc
def test_the_compiler_process_builds_and_reports_errors(tmp_path):
    """libv21.so never compiles inside the process that drives the GPU: it starts `v21_jitc` (csrc/jitc_main.cpp)."""
    exe = os.path.join(ROOT, "21cmvae_amd", "v21_jitc")
    assert os.path.exists(exe), "build() makes it next to libv21.so"
    d, err = str(tmp_path / "kc"), str(tmp_path / "e.txt")
    r = subprocess.run([exe, d, err, "1", "2", "7", "40", "451", "1", "0"], capture_output=True, text=True)
    assert r.returncode == 0 and len(os.listdir(d)) == 1 and not os.path.exists(err), r.stderr
    r = subprocess.run([exe, d, err, "1", "2", "7", "40", "451", "1", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "output layer is linear" in open(err).read()


def test_build_left_the_notebook_stacks_in_the_tree():
    """__graft_entry__.build() prebuilds the stacks the GPU tests and bench.py run at full size (kernel_cache/ travels to the
    GPU box with libv21.so), so the box loads code objects instead of compiling for minutes."""
    import __graft_entry__ as ge
    d = os.path.join(ROOT, "21cmvae_amd", "kernel_cache")
    if not os.path.isdir(d):
        pytest.skip("build() has not run in this tree")
    have = os.listdir(d)
    for dims, act in ge.PREBUILT_STACKS:
        spec = "x".join(map(str, dims)) + "_a" + "".join(str(a) for a in act)
        assert sum(f.startswith("fused_" + spec + "_Prec") for f in have) >= 3, spec
