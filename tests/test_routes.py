"""Which kernels a call takes ("no single table states which route a call takes by default, and no test asserts
it").  The table lives in INTEGRATION.md section 6; this file reads it from there.

CPU (`-m "not gpu"`): every row against the library's decision functions (csrc/routes.h through v21_route_forward /
v21_route_train: pure host logic), every environment switch the kernels' sources read is documented, and the switches move
the decision the way INTEGRATION.md section 5 says.

GPU (`-m gpu`): the rows marked `gpu` as REAL calls at their natural size with no V21_* switch set: the launch sites'
own record (v21_trainer_last_route / v21_mlp_last_route) must name the table's kernels, and the step's loss and FULL
gradient must meet the float64 oracle (tolerances of tests/helpers.py: f32 2e-5 / cos 0.999999, f16 3e-3 / 0.9995, bf16
3e-2 / 0.995) and a bitwise twin.  This is where the default large-step routes (fused_train16 from 8,193 rows, fused_train for
trainers of >= 24,576 rows, train_chain32 above 2,048 rows) run under the driver's suite at the sizes they ship for:
two workgroups per CU, several rounds, the XCD-major block order with more than one block per XCD, ragged last blocks."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, pkg
from helpers import STACKS, assert_step_matches_oracle, stack_data, twin_steps
from oracle import ref_numpy as ora


def _tables():
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = txt[txt.index("## 6. Route table"):txt.index("## 7. ")]
    train, fwd = [], []
    for line in sec.splitlines():
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) == 9 and c[0] in STACKS:
            train.append(dict(stack=c[0], prec=c[1], max_batch=int(c[2]), rows=int(c[3]), ranks=int(c[4]), rt=c[5] == "yes", fwd=c[6], upd=c[7], where=c[8]))
        elif len(c) == 6 and c[0] in STACKS:
            fwd.append(dict(stack=c[0], prec=c[1], rows=int(c[2]), rt=c[3] == "yes", route=c[4], where=c[5]))
    return train, fwd


TRAIN_ROWS, FWD_ROWS = _tables()
_tid = lambda r: "%s-%s-mb%d-rows%d-r%d%s" % (r["stack"], r["prec"], r["max_batch"], r["rows"], r["ranks"], "-rt" if r["rt"] else "")
_fid = lambda r: "%s-%s-rows%d-%s" % (r["stack"], r["prec"], r["rows"], "rt" if r["rt"] else "nort")
ROUTE_ENV = ["V21_TRAIN_CHAIN", "V21_FUSED_TRAIN", "V21_FUSED_TRAIN16", "V21_FUSED_TRAIN_ROWS", "V21_DW_SPLIT_ROWS", "V21_CHAIN32S",
             "V21_C32S_ROWS", "V21_DW32_LDS", "V21_DW32_ADAM", "V21_JIT", "V21_TRAIN_X16", "V21_SWEEP32_GROUP", "V21_CHAIN_PLAIN",
             "V21_DW_BLOCKS", "V21_CHAIN_PREF", "V21_DW_XROWS", "V21_SWEEP_STREAMS"]


@pytest.fixture
def no_switches(monkeypatch):
    for k in ROUTE_ENV:
        monkeypatch.delenv(k, raising=False)


def test_the_table_is_there():
    assert len(TRAIN_ROWS) >= 24 and len(FWD_ROWS) >= 15
    assert sum(r["where"] == "gpu" for r in TRAIN_ROWS) >= 12 and sum(r["where"] == "gpu" for r in FWD_ROWS) >= 6
    # every training kernel family and every forward route appears
    assert {r["fwd"] for r in TRAIN_ROWS} == {"per_layer", "chain16", "fused64", "fused128", "chain32", "chain32s_4", "chain32s_8"}
    assert {r["upd"] for r in TRAIN_ROWS} >= {"per_layer", "dw16_adam", "dw16_splitk", "dwadam32", "nt_sliced"}
    assert {r["route"] for r in FWD_ROWS} == {"small", "fused", "fused_rt", "table", "generic"}


@pytest.mark.parametrize("row", TRAIN_ROWS, ids=_tid)
def test_training_route_decision_equals_the_documented_table(row, no_switches):
    native = pkg("_native")
    dims, act = STACKS[row["stack"]]
    assert native.route_train(dims, act, row["prec"], row["max_batch"], row["rows"], row["ranks"], rt_ready=row["rt"]) == (row["fwd"], row["upd"])


@pytest.mark.parametrize("row", FWD_ROWS, ids=_fid)
def test_forward_route_decision_equals_the_documented_table(row, no_switches):
    native = pkg("_native")
    dims, act = STACKS[row["stack"]]
    assert native.route_forward(dims, act, row["prec"], row["rows"], rt_ready=row["rt"]) == row["route"]


def test_forward_flags_move_the_route(no_switches):
    native = pkg("_native")
    d1, nb = STACKS["D1"], STACKS["NB"]
    F = {"generic": 4, "no_small": 8, "chain": 16, "jit": 32, "tin": 1}
    assert native.route_forward(*d1, "f16", 65536, flags=F["generic"]) == "generic"
    assert native.route_forward(*d1, "f16", 65536, flags=F["chain"]) == "table"
    assert native.route_forward(*d1, "f16", 65536, flags=F["jit"], rt_ready=True) == "fused_rt"
    assert native.route_forward(*d1, "f32", 100, flags=F["no_small"]) == "fused"
    assert native.route_forward(*nb, "f32", 100, flags=F["no_small"], rt_ready=True) == "fused_rt"
    assert native.route_forward(*nb, "f32", 100, flags=F["no_small"]) == "table"
    # the fused parameter transform handles up to 8 input columns: a 451-wide input with the transform asked for is generic
    assert native.route_forward(*STACKS["AE"], "f16", 65536, flags=F["tin"]) == "generic"


def test_environment_switches_move_the_decision(monkeypatch, no_switches):
    native = pkg("_native")
    ae = STACKS["AE"]
    assert native.route_train(*ae, "f16", 777, 777) == ("chain16", "dw16_adam")
    monkeypatch.setenv("V21_FUSED_TRAIN_ROWS", "1")
    assert native.route_train(*ae, "f16", 777, 777) == ("fused64", "dw16_splitk")
    monkeypatch.setenv("V21_FUSED_TRAIN16", "0")
    assert native.route_train(*ae, "f16", 777, 777) == ("fused128", "dw16_splitk")
    monkeypatch.setenv("V21_FUSED_TRAIN", "0")
    assert native.route_train(*ae, "f16", 777, 777) == ("chain16", "dw16_adam")
    monkeypatch.setenv("V21_DW_SPLIT_ROWS", "512")
    assert native.route_train(*ae, "f16", 777, 777) == ("chain16", "dw16_splitk")
    monkeypatch.setenv("V21_TRAIN_CHAIN", "0")
    assert native.route_train(*ae, "f16", 777, 777) == ("per_layer", "per_layer")
    assert native.route_train(*ae, "f32", 256, 256) == ("per_layer", "per_layer")
    monkeypatch.delenv("V21_TRAIN_CHAIN")
    assert native.route_train(*ae, "f32", 256, 256) == ("chain32s_4", "dwadam32")
    monkeypatch.setenv("V21_C32S_ROWS", "8")
    assert native.route_train(*ae, "f32", 256, 256) == ("chain32s_8", "dwadam32")
    monkeypatch.setenv("V21_CHAIN32S", "0")
    assert native.route_train(*ae, "f32", 256, 256) == ("chain32", "dwadam32")
    monkeypatch.setenv("V21_DW32_LDS", "0")
    assert native.route_train(*ae, "f32", 256, 256) == ("chain32", "nt_dwadam")
    monkeypatch.setenv("V21_DW32_ADAM", "0")
    assert native.route_train(*ae, "f32", 256, 256) == ("chain32", "nt_sliced")


def test_every_environment_switch_the_library_reads_is_documented():
    """`grep getenv csrc/` against INTEGRATION.md section 5: a switch that ships undocumented is a finding."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = set()
    src_dir = os.path.join(ROOT, "21cmvae_amd", "csrc")
    for f in os.listdir(src_dir):
        if f.endswith((".hip", ".h", ".cpp")):
            names |= set(re.findall(r'(?:getenv|flag|num)\("(V21_[A-Z0-9_]+)"', open(os.path.join(src_dir, f)).read()))
    for f in os.listdir(os.path.join(ROOT, "21cmvae_amd")):
        if f.endswith(".py"):
            names |= set(re.findall(r'environ(?:\.get)?[\[(]"(V21_[A-Z0-9_]+)"', open(os.path.join(ROOT, "21cmvae_amd", f)).read()))
    assert len(names) >= 15
    missing = sorted(n for n in names if n not in doc)
    assert not missing, "undocumented switches: %s" % missing


def test_shipped_diagnostics_are_fenced_behind_diagnostic_builds():
    """V21_JIT_WIDE and V21_FUSED_DBG_PTR are read only inside #ifdef'ed diagnostic code."""
    src_dir = os.path.join(ROOT, "21cmvae_amd", "csrc")
    for f in os.listdir(src_dir):
        if not f.endswith((".hip", ".h")):
            continue
        depth_diag = []
        for line in open(os.path.join(src_dir, f)):
            t = line.strip()
            if t.startswith(("#ifdef", "#if ")):
                depth_diag.append("V21_DIAG" in t or "V21_FUSED_STAMP" in t)
            elif t.startswith("#ifndef"):
                depth_diag.append(False)
            elif t.startswith("#else") and depth_diag:
                depth_diag[-1] = False
            elif t.startswith("#endif") and depth_diag:
                depth_diag.pop()
            for name in ("V21_JIT_WIDE", "V21_FUSED_DBG_PTR"):
                if 'getenv("%s")' % name in line:
                    assert any(depth_diag), "%s is read by the product build (%s)" % (name, f)


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("row", [r for r in TRAIN_ROWS if r["where"] == "gpu"], ids=_tid)
def test_default_training_routes_at_natural_size_match_table_oracle_and_twin(ctx, row, no_switches):
    dims, act = STACKS[row["stack"]]
    n, rows, prec = row["max_batch"], row["rows"], row["prec"]
    case = TRAIN_ROWS.index(row)
    x, y, w = stack_data(dims, n, seed=100 + case)
    perm = np.random.default_rng(case).permutation(n).astype(np.int32) if case % 2 else None   # with and without an index table
    # first step at the row's size; then an epoch of two steps (the second partial) -- the fused kernels read a weight
    # stream written by the previous step's Adam pass from their second consecutive step on
    b2 = rows // 2 + 3 if rows > 1000 else None
    twins, weights = twin_steps(ctx, dims, act, prec, n, x, y, w, perm, rows, more=((perm, b2),), wait_jit=row["rt"])
    route = twins[0][3]
    assert route == (row["fwd"], row["upd"]), (row, route)
    assert_step_matches_oracle(_tid(row), twins, weights, act, x, y, w, perm, rows, prec)
    # ... and the decision the table test checks on the CPU is the one the launch sites followed
    native = pkg("_native")
    assert native.route_train(dims, act, prec, n, rows, 1, rt_ready=row["rt"]) == route


@pytest.mark.gpu
@pytest.mark.parametrize("row", [r for r in FWD_ROWS if r["where"] == "gpu"], ids=_fid)
def test_default_forward_routes_match_table_and_oracle(ctx, row, no_switches):
    native = pkg("_native")
    dims, act = STACKS[row["stack"]]
    prec, n = row["prec"], row["rows"]
    Ws, bs = ora.init_mlp(dims if 2 not in act else dims, seed=5)
    st = native.Stack(ctx, dims, act)
    if 2 in act:   # variational head: [z_mean | z_log_var] columns; predict evaluates z = z_mean (include/v21.h)
        gl = act.index(2)
        rng = np.random.default_rng(1)
        Wg = np.concatenate([Ws[gl], rng.normal(scale=0.01, size=Ws[gl].shape).astype(np.float32)], axis=1)
        bg = np.concatenate([bs[gl], np.zeros_like(bs[gl])])
        flat = ora.flatten_params(Ws[:gl] + [Wg] + Ws[gl + 1:], bs[:gl] + [bg] + bs[gl + 1:])
    else:
        flat = ora.flatten_params(Ws, bs)
    st.set_weights(flat)
    if row["rt"]:
        assert st.jit(prec, wait_ms=-1) == "ready"
    x = np.random.default_rng(2).uniform(-1, 1, size=(n, dims[0])).astype(np.float32)
    y = st.forward(x, prec)
    route, counts = st.last_route()
    assert route == row["route"], (row, route, counts)
    assert set(counts) == {row["route"]}, counts      # every slice of the host call took the same route
    assert native.route_forward(dims, act, prec, n, rt_ready=row["rt"]) == route
    pick = np.unique(np.concatenate([np.arange(min(n, 40)), np.random.default_rng(3).integers(0, n, 200), [n - 1]]))
    ref = x[pick].astype(np.float64)
    for W_, b_, a_ in zip(Ws, bs, act):                  # (a variational head evaluates z = z_mean: its layer is linear here)
        ref = ref @ W_.astype(np.float64) + b_.astype(np.float64)
        ref = np.maximum(ref, 0) if a_ == 1 else ref
    err = np.abs(y[pick] - ref)
    if prec == "f32":
        np.testing.assert_allclose(y[pick], ref, atol=2e-5, rtol=1e-5)
    else:
        lim = 1e-3 if prec == "f16" else 1e-2
        assert err.max() < lim * max(1.0, np.abs(ref).max()), (row, err.max())
