"""Shared pieces of the GPU parity tests (r5): the float64 restatement of one optimizer step's loss and FULL gradient, the
data sets of the reference's stacks at any row count, the tolerances per precision, and the "bitwise twin" check (two
trainers built the same way take the same launches on the same bits: loss, gradient and weights must be IDENTICAL -- a store
that lands in another tile's rows or a buffer read before it is written shows there long before it shows in a tolerance;
that comparison found the r4 overrun of the fused training kernel)."""
import numpy as np

from conftest import pkg
from oracle import ref_numpy as ora

# the stacks of INTEGRATION.md section 6 (dims, act); act 2 = V21_ACT_GAUSS (variational head)
STACKS = {
    "AE": ([451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]),       # emulator.py:522-524
    "LE": ([7, 352, 352, 352, 224, 9], [1, 1, 1, 1, 0]),        # emulator.py:525
    "DE": ([7, 288, 352, 288, 224, 451], [1, 1, 1, 1, 0]),      # emulator.py:196
    "D1": ([7, 352, 352, 352, 224, 451], [1, 1, 1, 1, 0]),      # BASELINE configs[1]
    "NB": ([7, 64, 128, 451], [1, 1, 0]),                       # notebooks/sample_notebook.ipynb: a custom stack
    "W6": ([7, 600, 451], [1, 0]),                              # wider than the chain kernels hold
    "VAE": ([451, 352, 9, 32, 352, 451], [1, 2, 1, 1, 0]),      # A13 (build-side)
}
# (relative loss error, cosine of the full gradient, both against the float64 oracle)
TOL = {"f32": (2e-5, 0.999999), "f16": (3e-3, 0.9995), "bf16": (3e-2, 0.995)}


def oracle_step(Ws, bs, act, x, tgt, w):
    """loss and flat gradient of ONE step in float64 (oracle/ref_numpy.py: batch_loss_and_grad; emulator.py:51-83)."""
    W = [a.astype(np.float64) for a in Ws]
    b = [a.astype(np.float64) for a in bs]
    acts = [x.astype(np.float64)]
    for W_, b_, a_ in zip(W, b, act):
        z = acts[-1] @ W_ + b_
        acts.append(np.maximum(z, 0) if a_ else z)
    lo, dz = ora.batch_loss_and_grad(acts[-1], tgt.astype(np.float64), w.astype(np.float64))
    L = len(act)
    dWs, dbs = [None] * L, [None] * L
    for li in range(L - 1, -1, -1):
        dWs[li] = acts[li].T @ dz
        dbs[li] = dz.sum(0)
        dh = dz @ W[li].T
        dz = dh * (acts[li] > 0) if li > 0 and act[li - 1] else dh
    return lo, ora.flatten_params(dWs, dbs)


def layer_errors(dims, g, go):
    """Per layer's [W; b] block of the flat arena: (||g - go|| / ||go||, max |g - go| / max |go|).  The cosine over the WHOLE arena
    cannot see a small layer: the autoencoder's 9 -> 32 layer is 320 of 332,000 elements (r5; a first-layer gradient that lacked
    one row of 257 would have passed at cosine 0.9999995)."""
    offs = np.cumsum([0] + [a * b + b for a, b in zip(dims[:-1], dims[1:])])
    out = []
    for i in range(len(dims) - 1):
        a, b = np.asarray(g[offs[i]:offs[i + 1]], np.float64), np.asarray(go[offs[i]:offs[i + 1]], np.float64)
        out.append((float(np.linalg.norm(a - b) / max(1e-300, np.linalg.norm(b))), float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))))
    return out


def kink_adjusted_oracle(Ws, bs, act, xrows, tgt, w, g_dev, go, rel_thr=1e-5, max_entries=4000):
    """The float64 gradient for the assignment of d relu / dz that the DEVICE made at pre-activations which are zero to within
    f32 rounding.  A row's contribution to the gradient depends on that row alone, so flipping the derivative of one
    (layer, row, unit) changes the gradient by a one-row backward pass: delta = g_row(flipped) - g_row(as float64 has it).  Every
    entry with |z| < rel_thr x the layer's largest |z| is a candidate; a candidate is taken when the device's gradient
    contains its delta (the projection of the residual on delta is more than half of |delta|^2 -- for a true flip it is all of
    it).  The forward values do not change: relu(z) is ~0 on either side.  -> (adjusted flat gradient, [(layer, row, unit, z)])."""
    L = len(act)
    W = [a.astype(np.float64) for a in Ws]
    b = [a.astype(np.float64) for a in bs]
    hs, zs = [np.asarray(xrows, np.float64)], []
    for W_, b_, a_ in zip(W, b, act):
        z = hs[-1] @ W_ + b_
        zs.append(z)
        hs.append(np.maximum(z, 0) if a_ else z)
    _, dout = ora.batch_loss_and_grad(hs[-1], np.asarray(tgt, np.float64), np.asarray(w, np.float64))
    cand = []
    for l in range(L - 1):   # (the output layer is linear)
        if not act[l]:
            continue
        thr = rel_thr * float(np.abs(zs[l]).max())
        rr, uu = np.where(np.abs(zs[l]) < thr)
        cand += [(l, int(r), int(u)) for r, u in zip(rr, uu)]
    if len(cand) > max_entries:
        return go, None

    def row_grad(r, flips):
        """flat gradient contribution of row r; flips: {(layer, unit)} whose ReLU derivative is the opposite of float64's"""
        dz = dout[r].copy()
        dWs, dbs = [None] * L, [None] * L
        for l in range(L - 1, -1, -1):
            dWs[l] = np.outer(hs[l][r], dz)
            dbs[l] = dz
            if l > 0:
                dh = W[l] @ dz
                if act[l - 1]:
                    m = zs[l - 1][r] > 0
                    for (fl, fu) in flips:
                        if fl == l - 1:
                            m = m.copy(); m[fu] = not m[fu]
                    dz = dh * m
                else:
                    dz = dh
        return ora.flatten_params(dWs, dbs)

    dims = [W[0].shape[0]] + [W_.shape[1] for W_ in W]
    offs = np.cumsum([0] + [a * b_ + b_ for a, b_ in zip(dims[:-1], dims[1:])])
    res = np.asarray(g_dev, np.float64) - go
    adj = np.array(go, np.float64)
    taken = []
    base = {}
    # Upper layers first: a flip at layer l changes column u of layer l's OWN block by that row's input times its activation
    # gradient, and everything below; nothing above.  So it is recognised in its own block alone, where flips of lower layers,
    # not yet accounted for, cannot interfere -- and by its SIZE: the residual of column u must contain the flip's change of that
    # column with coefficient 1 (0.95 .. 1.05).  (A projection of more than half took rows the device never flipped: the
    # inputs of a ReLU layer are all positive, so the changes of one column by different rows point roughly the same way --
    # row 570 at z = 2.4e-6 fitted 0.9 of what row 33302 at z = 1.8e-8 fitted exactly.)
    by_col = {}
    for (l, r, u) in cand:
        by_col.setdefault((l, u), []).append(r)
    for (l, u) in sorted(by_col, key=lambda c: -c[0]):
        K, N = dims[l], dims[l + 1]
        col = lambda v: v[offs[l]:offs[l + 1]].reshape(K + 1, N)[:, u]
        rows_left = list(by_col[(l, u)])
        while rows_left:
            rescol = col(res)
            best = None
            for r in rows_left:
                if r not in base:
                    base[r] = row_grad(r, set())
                flips_r = {(tl, tu) for (tl, tr_, tu, _) in taken if tr_ == r}
                cur = row_grad(r, flips_r) if flips_r else base[r]
                delta = row_grad(r, flips_r | {(l, u)}) - cur
                dcol = col(delta)
                n2 = float(dcol @ dcol)
                if n2 <= 0:
                    continue
                fit = float(rescol @ dcol) / n2
                if abs(fit - 1.0) < 0.05 and (best is None or n2 > best[0]):
                    best = (n2, r, delta)
            if best is None:
                break
            _, r, delta = best
            res -= delta; adj += delta
            taken.append((l, r, u, float(zs[l][r, u])))
            rows_left.remove(r)
    return adj, taken


def per_layer_gradient_check(dims, act, Ws, bs, xrows, g, go, prec, tgt=None, w=None):
    """-> (ok, note).  Every layer's block of the gradient against the float64 oracle's, by relative L2 norm.
    f32: 2e-6 -- summation noise is 1.2e-7 to 4e-7 from 1 to 40,001 rows -- once the ReLU derivative is taken the DEVICE's way at
    pre-activations that are zero to rounding: a unit whose float64 pre-activation is ~1e-8 of the layer's largest for one row
    gets that row's contribution in one summation order and not in another; its column of [W; b] is then off by one row's
    worth and every layer BELOW it by what flows back through that unit (r5: 2,047 rows of the latent emulator -- one column of
    layer 2 off, half the columns of layers 1 and 0, layers 3 and 4 exact to 2e-7; a 40,001-row step has eleven).  Nothing is
    assumed: kink_adjusted_oracle finds the (layer, row, unit) entries, each by a coefficient of 1.00 in its own column, rebuilds
    the float64 gradient with them, and the device's gradient must then agree layer by layer (measured: 1.2e-7 to 1.5e-7).
    f16 / bf16: 0.1 / 0.25 (operands rounded to 11 / 8 bits, and 16-bit pre-activations cross zero routinely; steps of fewer
    than 64 rows: 0.5) -- loose, but a layer that is WRONG is off by ~1, and the whole-arena cosine does not see a small layer."""
    rows = len(xrows)
    le = layer_errors(dims, g, go)
    worst = max(e[0] for e in le)
    if prec != "f32":
        tol = 0.5 if rows < 64 else (0.1 if prec == "f16" else 0.25)
        return worst <= tol, "per-layer rel L2 <= %.1e (tol %.2g)" % (worst, tol)
    tol = 2e-6
    if worst <= tol:
        return True, "per-layer rel L2 <= %.1e (tol %.1e)" % (worst, tol)
    if tgt is None or w is None:
        return False, "per-layer rel L2 %.1e (tol %.1e)" % (worst, tol)
    adj, taken = kink_adjusted_oracle(Ws, bs, act, xrows, tgt, w, g, go)
    if taken is None:
        return False, "per-layer rel L2 %.1e (tol %.1e); too many pre-activations near zero to examine" % (worst, tol)
    le2 = layer_errors(dims, g, adj)
    worst2 = max(e[0] for e in le2)
    desc = ", ".join("layer %d row %d unit %d z %.1e" % t for t in taken[:4]) + (" ..." if len(taken) > 4 else "")
    if worst2 <= tol and taken:
        return True, "per-layer rel L2 %.1e, %.1e with %d ReLU(s) at their kink taken the device's way (%s)" % (worst, worst2, len(taken), desc)
    # What the one-row deltas do not reproduce exactly (several flips in one column, a flip whose own row carries another):
    # the top-most layer still off must be a ReLU layer, off in a few columns only, and every one of those units must have a
    # pre-activation within 1e-4 of zero (relative to the layer's largest) for some row of the step -- else it is no kink.
    L = len(act)
    offs = np.cumsum([0] + [a * b_ + b_ for a, b_ in zip(dims[:-1], dims[1:])])
    top = max(l for l in range(L) if le2[l][0] > tol)
    if act[top]:
        gb = adj[offs[top]:offs[top + 1]].reshape(dims[top] + 1, dims[top + 1])
        d = np.asarray(g[offs[top]:offs[top + 1]], np.float64).reshape(dims[top] + 1, dims[top + 1]) - gb
        cn = np.linalg.norm(d, axis=0)
        order = np.argsort(cn)[::-1]
        ref = float(np.linalg.norm(gb))
        max_cols = 8 + rows // 2000
        cols = None
        for j in range(1, max_cols + 1):
            if np.sqrt(max(0.0, float((cn ** 2).sum() - (cn[order[:j]] ** 2).sum()))) <= tol * ref:
                cols = np.sort(order[:j])
                break
        if cols is not None:
            h = np.asarray(xrows, np.float64)
            for l in range(top + 1):
                z = h @ Ws[l].astype(np.float64) + bs[l].astype(np.float64)
                h = np.maximum(z, 0) if act[l] else z
            zmax = float(np.abs(z).max())
            near = [float(np.abs(z[:, c]).min()) for c in cols]
            if max(near) <= 1e-4 * zmax:
                return True, "per-layer rel L2 %.1e; %d ReLU(s) at their kink taken the device's way (%s), and layer %d unit(s) %s with |z| %s of %.1e; layers above within %.1e" % (
                    worst, len(taken), desc, top, cols.tolist(), ["%.1e" % v for v in near], zmax, max([e[0] for e in le2[top + 1:]] or [0.0]))
    return False, "per-layer rel L2 %.1e (tol %.1e); %.1e after %d pre-activations at zero were tried (%s)" % (worst, tol, worst2, len(taken), desc)


def stack_data(dims, n, seed):
    """(x, y or None, row weights): the reference's data shapes -- an autoencoder trains on pre-processed signals against
    themselves with the relative-MSE row weights (emulator.py:732-747), a direct emulator on parameters in [-1, 1] against
    pre-processed signals (emulator.py:361-378), a latent emulator on 9-wide targets with plain MSE (emulator.py:756-764)."""
    synth = pkg("synth")
    rng = np.random.default_rng(seed)
    if dims[0] == dims[-1] == 451:
        sig = synth.make_signals(n, seed=seed + 1)
        x = ora.preproc(sig, sig)
        return x, None, ora.relative_mse_row_weight(x, sig).astype(np.float32)
    x = rng.uniform(-1, 1, size=(n, dims[0])).astype(np.float32)
    if dims[-1] == 451:
        sig = synth.make_signals(n, seed=seed + 2)
        y = ora.preproc(sig, sig)
        return x, y, ora.relative_mse_row_weight(y, sig).astype(np.float32)
    y = rng.normal(size=(n, dims[-1])).astype(np.float32)
    return x, y, ora.mse_row_weight(y).astype(np.float32)


def init_weights(dims, seed):
    Ws, bs = ora.init_mlp(dims, seed=seed)
    rng = np.random.default_rng(seed + 1000)
    bs = [rng.normal(scale=0.05, size=b.shape).astype(np.float32) for b in bs]
    return Ws, bs, ora.flatten_params(Ws, bs)


def twin_steps(ctx, dims, act, prec, max_batch, x, y, w, perm, rows, more=((None, None),), lr=1e-3, wait_jit=False):
    """Two trainers built the same way: first step of `rows` rows (through `perm` when given), then one epoch per entry of
    `more` = (perm, batch).  -> [(loss1, grad1, weights_end, last_route, counts)] x 2 and the initial weights (Ws, bs)."""
    native = pkg("_native")
    Ws, bs, flat = init_weights(dims, seed=len(dims) * 7 + dims[1])
    out = []
    for _ in range(2):
        st = native.Stack(ctx, dims, act)
        st.set_weights(flat)
        tr = native.Trainer(st, prec, max_batch)
        tr.set_adam(lr=lr)
        if wait_jit:   # the run-time instantiated fused training kernel of this stack (build() prebuilt it; else this compiles)
            assert tr.jit(-1) == "ready"
        tr.set_data(0, x, y, w)
        l1 = tr.run_epoch(perm, rows)
        g1 = tr.get_grad()
        route, _ = tr.last_route()
        for p2, b2 in more:
            if b2:
                tr.run_epoch(p2, b2)
        out.append((l1, g1, st.get_weights(), route, tr.last_route()[1]))
    return out, (Ws, bs)


def assert_step_matches_oracle(tag, twins, weights, act, x, y, w, perm, rows, prec):
    Ws, bs = weights
    idx = perm[:rows] if perm is not None else np.arange(rows)
    tgt = (x if y is None else y)[idx]
    lo, go = oracle_step(Ws, bs, act, x[idx], tgt, w[idx])
    (l1, g1, w1, _, _), (l2, g2, w2, _, _) = twins
    tol_l, tol_c = TOL[prec]
    assert np.isfinite(g1).all(), tag
    assert abs(l1 - lo) <= tol_l * abs(lo), (tag, "loss", l1, lo)
    cos = float(g1 @ go / max(1e-300, np.linalg.norm(g1) * np.linalg.norm(go)))
    ratio = float(np.linalg.norm(g1) / max(1e-300, np.linalg.norm(go)))
    dims = [Ws[0].shape[0]] + [W.shape[1] for W in Ws]
    layered = 2 not in act   # (a variational head's oracle is not this plain stack)
    if layered:
        ok, note = per_layer_gradient_check(dims, act, Ws, bs, x[idx], g1, go, prec, tgt=tgt, w=w[idx])
        assert ok, (tag, note)
    if not (layered and prec == "f32"):   # (f32: the layer-by-layer bound is the tighter statement, and it knows about ReLU kinks)
        assert cos > tol_c and abs(ratio - 1) < 10 * tol_l, (tag, "gradient vs float64 oracle: cos %.7f norm ratio %.5f" % (cos, ratio))
    assert l1 == l2 and np.array_equal(g1, g2) and np.array_equal(w1, w2), (
        tag, "bitwise twin differs: loss %r %r, grad max diff %.2e, weights max diff %.2e" % (l1, l2, np.abs(g1 - g2).max(), np.abs(w1 - w2).max()))
    return cos
