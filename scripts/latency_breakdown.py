import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
emu = importlib.import_module("21cmvae_amd.emulator"); pp = importlib.import_module("21cmvae_amd.preprocess")
synth = importlib.import_module("21cmvae_amd.synth"); nat = importlib.import_module("21cmvae_amd._native")
data = synth.make_dataset(4000, 400, 400)
em = emu.DirectEmulator(hidden_dims=[352, 352, 352, 224], precision="f16", **data)
p1 = data["par_test"][0]
em.predict(p1)
def t(f, n=2000):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("predict total        %.1f us" % t(lambda: em.predict(p1)))
print("par_transform        %.1f us" % t(lambda: pp.par_transform(p1, em.par_train)))
x = pp.par_transform(p1, em.par_train)
st = em.emulator._ensure_stack()
print("_ensure_stack        %.1f us" % t(lambda: em.emulator._ensure_stack()))
print("SignalStats.of       %.1f us" % t(lambda: pp.SignalStats.of(em.signal_train)))
print("stack.forward f16    %.1f us" % t(lambda: st.forward(x, "f16", flags=nat.FWD_OUT_TRANSFORM)))
print("stack.forward f32    %.1f us" % t(lambda: st.forward(x, "f32", flags=nat.FWD_OUT_TRANSFORM)))
x32 = x.astype(np.float32)
print("stack.forward f16/x32 %.1f us" % t(lambda: st.forward(x32, "f16", flags=nat.FWD_OUT_TRANSFORM)))
ctx = st.ctx
d_x, d_y = ctx.malloc(64), ctx.malloc(451 * 4)
def dev():
    st.forward_dev(d_x, 7, 1, d_y, 451, "f16", nat.FWD_OUT_TRANSFORM); ctx.sync()
print("forward_dev+sync f16 %.1f us" % t(dev))
def dev32():
    st.forward_dev(d_x, 7, 1, d_y, 451, "f32", nat.FWD_OUT_TRANSFORM); ctx.sync()
print("forward_dev+sync f32 %.1f us" % t(dev32))
