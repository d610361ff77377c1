"""Poison tests (SURVEY section 5, row "race detection / sanitizers"; GPU AddressSanitizer is not available on
this pool, so the checks are done with data): every kernel that keeps activations, masks or partial tiles in
LDS must produce the same bits whether the LDS it finds holds a fresh process's zeros or NaN patterns left by
another kernel, and a forward launch must write rows < n only.

(i)  ``Context.poison_lds`` (v21_debug_poison_lds) fills all 160 KB of every CU's LDS with 0xFFFFFFFF -- NaN as
     fp32, f16 and bf16 -- immediately before the launch under test; the result must equal the unpoisoned run BIT FOR
     BIT and meet the oracle.  Shapes are ragged on purpose: a 7-wide input (K padded to 16), a 9-wide latent, a
     451-wide output, batches that are not multiples of 32.
(ii) the output buffer (n + pad rows) is pre-filled with NaN: rows < n must come back finite, rows >= n untouched.
"""
import numpy as np
import pytest

from conftest import pkg
from oracle import ref_numpy as ora

pytestmark = pytest.mark.gpu

AE_DIMS, AE_ACT = [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]
EM_DIMS, EM_ACT = [7, 352, 352, 352, 224, 9], [1, 1, 1, 1, 0]


def _trainer(ctx, dims, act, prec, batch, seed, lr=1e-3):
    native = pkg("_native")
    Ws, bs = ora.init_mlp(dims, seed=seed)
    st = native.Stack(ctx, dims, act)
    st.set_weights(ora.flatten_params(Ws, bs))
    tr = native.Trainer(st, prec, batch)
    tr.set_adam(lr=lr)
    return st, tr, Ws, bs


def _oracle_loss(Ws, bs, act, x, y, w):
    h = x.astype(np.float64)
    for W_, b_, a_ in zip(Ws, bs, act):
        h = h @ W_.astype(np.float64) + b_.astype(np.float64)
        h = np.maximum(h, 0) if a_ else h
    return float(np.mean(ora.per_sample_loss(h, y.astype(np.float64), w.astype(np.float64))))


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("case", ["autoencoder_451", "emulator_7_to_9", "direct_7_to_451"])
def test_chain_step_does_not_depend_on_what_lds_held(ctx, prec, case):
    """train_chain_kernel + dw16_adam_kernel (one optimizer step, then a forward-only validation launch) after a
    NaN fill of every CU's LDS: bit-identical to the clean run, finite, and the loss meets the float64 oracle."""
    synth = pkg("synth")
    n = 200 + 13  # 6 full row blocks + 21 rows
    rng = np.random.default_rng(3)
    if case == "autoencoder_451":
        dims, act = AE_DIMS, AE_ACT
        sig = synth.make_signals(n, seed=5)
        x = ora.preproc(sig, sig); y = None
        w = ora.relative_mse_row_weight(x, sig).astype(np.float32)
    elif case == "emulator_7_to_9":
        dims, act = EM_DIMS, EM_ACT
        x = rng.uniform(-1, 1, size=(n, 7)).astype(np.float32)
        y = rng.normal(size=(n, 9)).astype(np.float32)
        w = ora.mse_row_weight(y).astype(np.float32)
    else:
        dims, act = [7, 288, 352, 288, 224, 451], [1, 1, 1, 1, 0]
        par = synth.make_params(n, seed=3)
        x = ora.par_transform(par, par).astype(np.float32)
        sig = synth.signals_from_params(par)
        y = ora.preproc(sig, sig)
        w = ora.relative_mse_row_weight(y, sig).astype(np.float32)
    res = {}
    for poison in (False, True):
        st, tr, Ws, bs = _trainer(ctx, dims, act, prec, n, seed=17)
        tr.set_data(0, x, y, w)
        tr.set_data(1, x[:77], None if y is None else y[:77], w[:77])
        out = []
        for _ in range(2):
            if poison:
                ctx.poison_lds()
            out.append(tr.run_epoch(None, n))  # ONE chain launch + one gradient/Adam launch
        if poison:
            ctx.poison_lds()
        out.append(tr.evaluate(1, 77))         # the forward-only launch
        res[poison] = (np.array(out), tr.get_grad(), st.get_weights())
    for a, b in zip(res[True], res[False]):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)
    lo = _oracle_loss(Ws, bs, act, x, x if y is None else y, w)
    tol = 3e-3 if prec == "f16" else 3e-2
    assert abs(res[True][0][0] - lo) / lo < tol, (res[True][0][0], lo)


def test_variational_chain_step_after_lds_poison(ctx):
    """The variational head keeps (z_mean | z_log_var) and KL_i in LDS as fp32 (train_chain.h: zs, klb)."""
    native, synth = pkg("_native"), pkg("synth")
    dims, act = [451, 96, 18, 32, 451], [1, native.ACT_GAUSS, 1, 0]
    dims_dense = [451, 96, 9, 32, 451]
    n = 75
    sig = synth.make_signals(n, seed=8)
    x = ora.preproc(sig, sig)
    w = ora.relative_mse_row_weight(x, sig).astype(np.float32)
    res = {}
    for poison in (False, True):
        st = native.Stack(ctx, dims_dense, act)
        st.set_weights((np.random.default_rng(2).normal(size=st.num_params) * 0.05).astype(np.float32))
        tr = native.Trainer(st, "f16", n)
        tr.set_adam(lr=1e-3)
        tr.set_vae(1e-3, sample=True, seed=5)
        tr.set_data(0, x, None, w)
        out = []
        for _ in range(2):
            if poison:
                ctx.poison_lds()
            out.append(tr.run_epoch(None, n))
        res[poison] = (np.array(out), tr.get_grad(), st.get_weights())
    for a, b in zip(res[True], res[False]):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)


def test_joint_and_grouped_chain_after_lds_poison(ctx):
    """train_chain_joint_kernel (two models through one workgroup, latents captured in LDS) and the grouped chain
    of a sweep, ragged batch, full-width autoencoder + latent emulator."""
    native, synth = pkg("_native"), pkg("synth")
    n = 150
    sig = synth.make_signals(n, seed=6)
    y = ora.preproc(sig, sig)
    par = np.random.default_rng(9).uniform(-1, 1, size=(n, 7)).astype(np.float32)
    wa = ora.relative_mse_row_weight(y, sig).astype(np.float32)
    wz = ora.mse_row_weight(np.zeros((n, 9), np.float32)).astype(np.float32)
    res = {}
    for poison in (False, True):
        sta, tra, _, _ = _trainer(ctx, AE_DIMS, AE_ACT, "f16", n, seed=41)
        ste, tre, _, _ = _trainer(ctx, EM_DIMS, EM_ACT, "f16", n, seed=42)
        tra.set_data(0, y, None, wa)
        tre.set_data(0, par, np.zeros((n, 9), np.float32), wz)
        joint = native.Joint(tra, tre, latent_layer=1)
        out = []
        for _ in range(2):
            if poison:
                ctx.poison_lds()
            out.extend(joint.run_epoch(None, n))
        # a sweep of two autoencoders that differ in width (grouped chain + grouped gradient/Adam launch)
        trs = []
        for k, dims in enumerate(([451, 128, 4, 32, 128, 451], [451, 352, 9, 32, 352, 451])):
            _, t, _, _ = _trainer(ctx, dims, AE_ACT, "f16", n, seed=50 + k)
            trs.append(t)
        trs[0].set_data(0, y, None, wa)
        sw = native.Sweep(trs)
        for _ in range(2):
            if poison:
                ctx.poison_lds()
            out.extend(sw.run_epoch(None, n))
        res[poison] = (np.array(out), sta.get_weights(), ste.get_weights(), trs[0].stack.get_weights(), trs[1].stack.get_weights())
    for a, b in zip(res[True], res[False]):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("prec", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("n", [1, 31, 257, 65553])
def test_forward_writes_rows_below_n_only_and_ignores_lds_content(ctx, prec, n):
    """fused_fwd on the headline stack (7-wide input, 451-wide output): the output buffer holds n + 40 rows of NaN
    before the launch, LDS holds NaN; afterwards rows < n are finite and equal the clean run, rows >= n still hold
    the fill pattern."""
    native = pkg("_native")
    dims, act = [7, 352, 352, 352, 224, 451], [1, 1, 1, 1, 0]
    Ws, bs = ora.init_mlp(dims, seed=3)
    st = native.Stack(ctx, dims, act)
    st.set_weights(ora.flatten_params(Ws, bs))
    pad = 40
    x = np.random.default_rng(n).uniform(-1, 1, size=(n, 7)).astype(np.float32)
    d_x = ctx.malloc(x.nbytes)
    d_y = ctx.malloc((n + pad) * 451 * 4)
    ctx.h2d(d_x, x)
    outs = []
    for poison in (False, True):
        ctx.memset(d_y, 0xFF, (n + pad) * 451 * 4)
        if poison:
            ctx.poison_lds()
        st.forward_dev(d_x, 7, n, d_y, 451, prec, native.FWD_NO_SMALL)
        y = np.empty((n + pad, 451), np.float32)
        ctx.d2h(y, d_y)
        assert np.isfinite(y[:n]).all()
        assert (y[n:].view(np.uint32) == 0xFFFFFFFF).all(), "rows past n were written"
        outs.append(y[:n].copy())
    np.testing.assert_array_equal(outs[0], outs[1])
    pick = np.unique(np.r_[0, n - 1, np.random.default_rng(1).integers(0, n, size=20)])
    ref = ora.mlp_forward(Ws, bs, x[pick])
    if prec == "f32":
        np.testing.assert_allclose(outs[1][pick], ref, atol=2e-5, rtol=1e-5)
    else:
        assert np.abs(outs[1][pick] - ref).max() <= (1e-3 if prec == "f16" else 1e-2)
    ctx.free(d_x); ctx.free(d_y)


def test_small_batch_and_generic_forward_after_lds_poison(ctx):
    """The per-layer paths (gemm_nt small-batch route, gemm.h K-loop route of stacks without a fused kernel)."""
    native = pkg("_native")
    dims, act = [7, 64, 128, 451], [1, 1, 0]   # notebooks/sample_notebook.ipynb's custom model: no fused kernel
    Ws, bs = ora.init_mlp(dims, seed=5)
    st = native.Stack(ctx, dims, act)
    st.set_weights(ora.flatten_params(Ws, bs))
    for n in (1, 33, 5000):
        x = np.random.default_rng(n).uniform(-1, 1, size=(n, 7)).astype(np.float32)
        ref = ora.mlp_forward(Ws, bs, x)
        for prec in ("f32", "f16"):
            clean = st.forward(x, prec)
            ctx.poison_lds()
            y = st.forward(x, prec)
            np.testing.assert_array_equal(y, clean)
            if prec == "f32":
                np.testing.assert_allclose(y, ref, atol=2e-5, rtol=1e-5)
            else:
                assert np.abs(y - ref).max() <= 2e-3
