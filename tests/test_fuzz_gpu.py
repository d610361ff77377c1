"""A seeded slice of every fuzzer of scripts/diag/ under `pytest -m gpu` ("promote the fuzzers into
-m gpu"; the r4 overrun of the fused training kernel was found by train_fuzz.py, not by pytest).  The scripts hold the
case generators and the checks (gen_cases / run_case); here every case is one parametrised test with a fixed seed, so a
failure names its case and `FUZZ_ONLY=<case> python scripts/diag/<name>_fuzz.py <cases> <seed>` reproduces it.

train     loss + FULL gradient of a random step against the float64 oracle, and a bitwise twin
train_big large steps of the reference stacks on the library's own route choice (FUZZ_BIG)
forward   random forward calls (route, transforms, dtypes, host / strided device buffers) against the float64 oracle, twice
joint     joint steps through two invariants that need no second implementation + a twin
sweep     grouped steps against the members trained one by one + a twin
surface   the class surface end to end (train, predict in float32 / float64, save -> load, test_error)
dp        2-4 ranks on the ONE GPU (gloo + the host-staged transport) against one process; ranks bit-identical"""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _load(name):
    # (a regular import by module name: dp_fuzz starts its ranks with multiprocessing's spawn, which pickles the worker
    #  function by module name and hands the children this process's sys.path)
    d = os.path.join(ROOT, "scripts", "diag")
    if d not in sys.path:
        sys.path.insert(0, d)
    return importlib.import_module(name + "_fuzz")


train_fuzz, forward_fuzz, joint_fuzz, sweep_fuzz, surface_fuzz, dp_fuzz = (_load(n) for n in ("train", "forward", "joint", "sweep", "surface", "dp"))
ROUTE_ENV = ["V21_TRAIN_CHAIN", "V21_FUSED_TRAIN", "V21_FUSED_TRAIN16", "V21_FUSED_TRAIN_ROWS", "V21_DW_SPLIT_ROWS", "V21_CHAIN32S", "V21_C32S_ROWS"]


@pytest.fixture
def clean_env(monkeypatch):
    for k in ROUTE_ENV:
        monkeypatch.delenv(k, raising=False)
    return monkeypatch


def _check(status, msg, tag):
    assert status in ("OK", "refused"), "%s\n%s" % (tag, msg)


@pytest.mark.parametrize("k", list(train_fuzz.gen_cases(12, seed=11)), ids=lambda k: "c%d-%s-rows%d" % (k["c"], k["prec"], k["rows"]))
def test_train_fuzz_slice(ctx, k, clean_env):
    _check(*train_fuzz.run_case(ctx, k, setenv=clean_env.setenv, delenv=lambda n: clean_env.delenv(n, raising=False)), train_fuzz.tag_of(k))


@pytest.mark.parametrize("k", list(train_fuzz.gen_cases(6, seed=12, big=True)), ids=lambda k: "c%d-%s-rows%d" % (k["c"], k["prec"], k["rows"]))
def test_train_fuzz_slice_of_large_steps_on_the_default_routes(ctx, k, clean_env):
    _check(*train_fuzz.run_case(ctx, k, setenv=clean_env.setenv, delenv=lambda n: clean_env.delenv(n, raising=False)), train_fuzz.tag_of(k))


@pytest.mark.parametrize("k", list(forward_fuzz.gen_cases(14, seed=13, families_only=True)), ids=lambda k: "c%d-%s-n%d-%s" % (k["c"], k["prec"], k["n"], k["rname"]))
def test_forward_fuzz_slice(ctx, k, clean_env):
    _check(*forward_fuzz.run_case(ctx, k), forward_fuzz.tag_of(k))


@pytest.mark.parametrize("k", list(joint_fuzz.gen_cases(10, seed=14)), ids=lambda k: "c%d-%s-n%d-b%d" % (k["c"], k["prec"], k["n"], k["batch"]))
def test_joint_fuzz_slice(ctx, k, clean_env):
    _check(*joint_fuzz.run_case(ctx, k), joint_fuzz.tag_of(k))


def test_f32_joint_drift_against_the_separate_trainer_grows_with_the_step_count(clean_env):
    """joint_fuzz case 115 (seed 105: f32, autoencoder 33-352-288-16-16-512-33 frozen, emulator 7-400-16, 1,500
    single-row steps per epoch) came out 5.4e-5 off the separate trainer on the oracle's latents, and r4 answered by widening the
    tolerance to 1e-4 with "accumulated rounding" asserted, not shown.  Shown here, on the very case (scripts/diag/joint_case115_r4.py
    keeps r4's generator so that it can be drawn again), epoch by epoch over four epochs: the two runs agree to ~6e-9 after the
    first 1,500 steps and then SEPARATE -- 5e-5, 7e-5, 3e-4 -- until the emulator's weights differ by more than half their
    range: two fp32 trajectories of batch-1 Adam steps (m / (sqrt(v) + eps) turns a 1e-8 difference of the targets -- fp32
    latents formed on the device against float32 roundings of the float64 oracle's -- into different updates wherever a
    gradient is small), not an error per step: fresh draws of the same shapes stay at 1e-8 for 3,000 steps (r5 runs).  The
    invariant is therefore checked over TWO epochs, at 2e-5 for epochs of up to 64 steps and 1e-4 beyond (joint_fuzz.py)."""
    import subprocess
    env = dict(os.environ, FUZZ_ONLY="115")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "diag", "joint_case115_r4.py"), "200", "105"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("relative difference by epoch")]
    assert line, r.stdout[-2000:]
    rel = [float(v) for v in line[0].split("[")[1].rstrip("]").split(",")]
    print("\n" + "\n".join(ln for ln in r.stdout.splitlines() if ln.startswith(("epoch losses", "relative difference", "emulator weights"))))
    assert len(rel) == 4 and rel[0] < 1e-6                    # 1,500 steps: the same run to rounding
    assert rel[1] < 1e-4                                      # the two epochs joint_fuzz.py compares: inside its bound
    assert rel[3] > 10 * rel[0] and max(rel) < 5e-3           # ... and it GROWS with the steps: trajectories separating


@pytest.mark.parametrize("k", list(sweep_fuzz.gen_cases(10, seed=21, max_count=64)), ids=lambda k: "c%d-%s-m%d-b%d" % (k["c"], k["prec"], k["count"], k["batch"]))
def test_sweep_fuzz_slice(ctx, k, clean_env):
    _check(*sweep_fuzz.run_case(ctx, k), sweep_fuzz.tag_of(k))


@pytest.mark.parametrize("k", list(surface_fuzz.gen_cases(8, seed=16)), ids=lambda k: "c%d-%s-%s-n%d-b%d" % (k["c"], k["kind"], k["prec"], k["n_train"], k["batch"]))
def test_surface_fuzz_slice(ctx, k, clean_env):
    _check(*surface_fuzz.run_case(k), surface_fuzz.tag_of(k))


@pytest.mark.parametrize("cfg", list(dp_fuzz.gen_cases(4, seed=17)), ids=lambda c: "c%d-w%d-%s-%s" % (c["c"], c["world"], c["prec"], "sharded" if c["sharded"] else "allreduce"))
def test_dp_fuzz_slice(ctx, cfg, clean_env):
    _check(*dp_fuzz.run_case(cfg), dp_fuzz.tag_of(cfg))
