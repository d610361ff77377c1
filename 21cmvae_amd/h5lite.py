"""Minimal read-only HDF5 reader + Keras legacy-H5 model loader (pure Python + numpy).

Why: the reference loads its weights with ``tf.keras.models.load_model`` (emulator.py:335,
:691-693) and its dataset with h5py (emulator.py:198-204); neither TensorFlow nor h5py
is available where this engine runs.  The files involved are simple -- HDF5 "earliest"
format as written by h5py/Keras 2.7: superblock v0/v1, old-style groups (v1 B-tree +
symbol-table nodes + local heap), v1 object headers, contiguous (or compact)
un-filtered datasets of fixed-point / IEEE-float / fixed-length-string type, attributes
with fixed- or variable-length strings (global heap) -- and that subset is what this
module parses.  Anything else (chunked/compressed data, new-style groups) raises
``H5Unsupported`` rather than guessing.  Nothing in a file is ever executed.

Layout reference: the public "HDF5 File Format Specification Version 2.0".
"""
import json
import os
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Unsupported(IOError):
    pass


class _Reader:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        sig = b"\x89HDF\r\n\x1a\n"
        if self.buf[:8] != sig:
            raise IOError("%s is not an HDF5 file" % path)
        ver = self.buf[8]
        if ver not in (0, 1):
            raise H5Unsupported("superblock version %d (only 0/1: files written with libver='earliest')" % ver)
        self.so, self.sl = self.buf[13], self.buf[14]  # size of offsets / lengths
        if self.so != 8 or self.sl != 8:
            raise H5Unsupported("offset/length sizes %d/%d" % (self.so, self.sl))
        p = 24 if ver == 0 else 28
        self.base = self.u64(p)
        # root group symbol table entry follows base, free-space, EOF, driver addresses
        self.root_entry = p + 32
        self.gheap_cache = {}

    def u8(self, p): return self.buf[p]
    def u16(self, p): return struct.unpack_from("<H", self.buf, p)[0]
    def u32(self, p): return struct.unpack_from("<I", self.buf, p)[0]
    def u64(self, p): return struct.unpack_from("<Q", self.buf, p)[0]

    # -- object headers (version 1) -------------------------------------------------
    def messages(self, addr):
        """Yield (type, flags, body_offset, size) for every header message of an object."""
        if self.buf[addr:addr + 4] == b"OHDR":
            raise H5Unsupported("version-2 object headers (file not written with libver='earliest')")
        if self.u8(addr) != 1:
            raise H5Unsupported("object header version %d" % self.u8(addr))
        nmsg = self.u16(addr + 2)
        size = self.u32(addr + 8)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self.u16(p), self.u16(p + 2), self.u8(p + 4)
                body = p + 8
                if mtype == 0x0010:  # continuation
                    blocks.append((self.u64(body), self.u64(body + 8)))
                out.append((mtype, flags, body, msize))
                p = body + msize
        return out

    # -- datatypes ---------------------------------------------------------------------
    def datatype(self, p):
        """-> (numpy dtype or ('vlen_str',), size_in_bytes, message_length)"""
        cv = self.u8(p)
        cls, ver = cv & 0x0F, cv >> 4
        b0 = self.u8(p + 1)
        size = self.u32(p + 4)
        if cls == 0:  # fixed point
            signed = (b0 >> 3) & 1
            order = ">" if b0 & 1 else "<"
            return np.dtype("%s%s%d" % (order, "i" if signed else "u", size)), size, 8 + 4
        if cls == 1:  # floating point
            order = ">" if b0 & 1 else "<"
            return np.dtype("%sf%d" % (order, size)), size, 8 + 12
        if cls == 3:  # fixed-length string
            return np.dtype("S%d" % size), size, 8
        if cls == 9:  # variable length
            is_str = (b0 & 0x0F) == 1
            if not is_str:
                raise H5Unsupported("variable-length sequences")
            return ("vlen_str",), size, None
        raise H5Unsupported("datatype class %d" % cls)

    def dataspace(self, p):
        ver = self.u8(p)
        rank = self.u8(p + 1)
        if ver == 1:
            q = p + 8
        elif ver == 2:
            q = p + 4
        else:
            raise H5Unsupported("dataspace version %d" % ver)
        return tuple(self.u64(q + 8 * i) for i in range(rank))

    def global_heap_object(self, addr, index):
        if addr not in self.gheap_cache:
            if self.buf[addr:addr + 4] != b"GCOL":
                raise IOError("bad global heap at %d" % addr)
            size = self.u64(addr + 8)
            objs, p, end = {}, addr + 16, addr + size
            while p + 16 <= end:
                idx, osize = self.u16(p), self.u64(p + 8)
                if idx == 0:
                    break
                objs[idx] = (p + 16, osize)
                p += 16 + ((osize + 7) // 8) * 8
            self.gheap_cache[addr] = objs
        q, n = self.gheap_cache[addr][index]
        return self.buf[q:q + n]

    def read_values(self, dt, shape, data_off):
        n = int(np.prod(shape)) if shape else 1
        if isinstance(dt, tuple):  # vlen strings: (length u32, heap addr u64, index u32) each
            out = []
            for i in range(n):
                q = data_off + 16 * i
                ln, ha, ix = self.u32(q), self.u64(q + 4), self.u32(q + 12)
                out.append(self.global_heap_object(ha, ix)[:ln].decode("utf-8") if ha not in (0, UNDEF) else "")
            return out[0] if not shape else np.array(out, dtype=object).reshape(shape)
        arr = np.frombuffer(self.buf, dtype=dt, count=n, offset=data_off)
        return arr.reshape(shape).copy() if shape else arr.reshape(()).copy()

    def attributes(self, addr):
        out = {}
        for mtype, _f, body, _s in self.messages(addr):
            if mtype != 0x000C:
                continue
            ver = self.u8(body)
            nsz, dsz, ssz = self.u16(body + 2), self.u16(body + 4), self.u16(body + 6)
            p = body + 8 + (1 if ver == 3 else 0)
            pad = (lambda v: (v + 7) // 8 * 8) if ver == 1 else (lambda v: v)
            name = self.buf[p:p + nsz].split(b"\0")[0].decode("utf-8")
            p += pad(nsz)
            dt, _, _ = self.datatype(p)
            p += pad(dsz)
            shape = self.dataspace(p) if ssz >= 2 and self.u8(p + 1) > 0 else ()
            p += pad(ssz)
            out[name] = self.read_values(dt, shape, p)
        return out

    # -- groups --------------------------------------------------------------------------
    def group_entries(self, addr):
        """name -> object header address, for an old-style group."""
        btree = heap = None
        for mtype, _f, body, _s in self.messages(addr):
            if mtype == 0x0011:
                btree, heap = self.u64(body), self.u64(body + 8)
            if mtype in (0x0002, 0x0006):
                raise H5Unsupported("new-style (link message) groups")
        if btree is None:
            return None
        if self.buf[heap:heap + 4] != b"HEAP":
            raise IOError("bad local heap")
        heap_data = self.u64(heap + 24)
        out = {}

        def walk(node):
            if self.buf[node:node + 4] == b"TREE":
                level, used = self.u8(node + 5), self.u16(node + 6)
                p = node + 24
                for i in range(used):
                    child = self.u64(p + 8 + i * 16)
                    walk(child)
            elif self.buf[node:node + 4] == b"SNOD":
                nsym = self.u16(node + 6)
                for i in range(nsym):
                    e = node + 8 + 40 * i
                    noff, ohdr = self.u64(e), self.u64(e + 8)
                    q = heap_data + noff
                    name = self.buf[q:self.buf.index(b"\0", q)].decode("utf-8")
                    out[name] = ohdr
            else:
                raise IOError("bad group B-tree node at %d" % node)

        walk(btree)
        return out

    def dataset(self, addr):
        dt = shape = None
        data = None
        for mtype, _f, body, _s in self.messages(addr):
            if mtype == 0x0003:
                dt, _, _ = self.datatype(body)
            elif mtype == 0x0001:
                shape = self.dataspace(body)
            elif mtype == 0x000B:
                raise H5Unsupported("filtered (compressed) datasets")
            elif mtype == 0x0008:
                ver = self.u8(body)
                if ver == 3:
                    cls = self.u8(body + 1)
                    if cls == 1:
                        data = ("contig", self.u64(body + 2), self.u64(body + 10))
                    elif cls == 0:
                        data = ("compact", body + 4, self.u16(body + 2))
                    else:
                        raise H5Unsupported("chunked datasets")
                elif ver in (1, 2):
                    rank, cls = self.u8(body + 1), self.u8(body + 2)
                    if cls != 1:
                        raise H5Unsupported("layout class %d (layout message v%d)" % (cls, ver))
                    data = ("contig", self.u64(body + 8), None)
                else:
                    raise H5Unsupported("data layout version %d" % ver)
        if dt is None or shape is None:
            return None
        return dt, shape, data


class Dataset:
    def __init__(self, rd, addr, info):
        self._rd, self._addr = rd, addr
        self.dtype, self.shape, self._data = info
        self.attrs = _LazyAttrs(rd, addr)

    def __getitem__(self, key):
        kind, off, _ = self._data if self._data else (None, None, None)
        if kind is None or off == UNDEF:
            arr = np.zeros(self.shape, self.dtype if not isinstance(self.dtype, tuple) else object)
        else:
            arr = self._rd.read_values(self.dtype, self.shape, off + (self._rd.base if kind == "contig" else 0))
        arr = np.asarray(arr)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        return arr[key] if arr.shape else arr[()]


class _LazyAttrs(dict):
    def __init__(self, rd, addr):
        super().__init__()
        self._rd, self._addr, self._loaded = rd, addr, False

    def _load(self):
        if not self._loaded:
            super().update(self._rd.attributes(self._addr))
            self._loaded = True

    def __getitem__(self, k):
        self._load(); return super().__getitem__(k)

    def __contains__(self, k):
        self._load(); return super().__contains__(k)

    def keys(self):
        self._load(); return super().keys()

    def items(self):
        self._load(); return super().items()

    def get(self, k, d=None):
        self._load(); return super().get(k, d)


class Group:
    def __init__(self, rd, addr, entries):
        self._rd, self._addr, self._entries = rd, addr, entries
        self.attrs = _LazyAttrs(rd, addr)

    def keys(self):
        return self._entries.keys()

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._entries:
                raise KeyError(path)
            addr = node._entries[part] + node._rd.base
            ent = node._rd.group_entries(addr)
            if ent is not None:
                node = Group(node._rd, addr, ent)
            else:
                info = node._rd.dataset(addr)
                if info is None:
                    raise KeyError(path)
                node = Dataset(node._rd, addr, info)
        return node


class File(Group):
    """``with File(path) as hf: hf["signal_train"][:]`` -- the h5py calls the reference makes."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise H5Unsupported("read-only")
        rd = _Reader(path)
        root = rd.u64(rd.root_entry + 8) + rd.base
        ent = rd.group_entries(root)
        if ent is None:
            raise H5Unsupported("root group is not an old-style group")
        super().__init__(rd, root, ent)
        self.filename = path

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ---- Keras legacy-H5 model files ------------------------------------------------------
def _s(v):
    if isinstance(v, np.ndarray):  # a scalar (0-d or 1-element) fixed-length string attribute
        v = v.reshape(-1)[0]
    return v.decode("utf-8") if isinstance(v, (bytes, np.bytes_)) else str(v)


def read_keras_h5(path):
    """-> dict(layers=[(name, kernel, bias, activation)], optimizer=... or None, config=...)
    Format: root attrs model_config/training_config (JSON), group
    model_weights with attr layer_names; per layer attr weight_names -> datasets
    <layer>/<layer>/kernel:0 (in,out) f32 and bias:0; optional optimizer_weights/Adam/*."""
    with File(path) as f:
        if "model_weights" not in f:
            raise IOError("%s holds no Keras model (no model_weights group)" % path)
        cfg = None
        try:
            mc = f.attrs.get("model_config")
            cfg = json.loads(_s(mc)) if mc is not None else None
        except (H5Unsupported, ValueError):
            cfg = None
        acts = {}
        if cfg:
            layer_cfgs = cfg.get("config", {}).get("layers", []) if isinstance(cfg.get("config"), dict) else cfg.get("config", [])
            for lc in layer_cfgs:
                if lc.get("class_name") == "Dense":
                    acts[lc["config"]["name"]] = lc["config"].get("activation", "linear")
        mw = f["model_weights"]
        names = [_s(n) for n in np.atleast_1d(mw.attrs["layer_names"])]
        layers = []
        for name in names:
            g = mw[name]
            wn = [_s(w) for w in np.atleast_1d(g.attrs["weight_names"])] if "weight_names" in g.attrs else []
            if not wn:
                continue
            kern = np.asarray(g[wn[0]][:], np.float32)
            bias = np.asarray(g[wn[1]][:], np.float32)
            layers.append([name, kern, bias, acts.get(name)])
        for i, l in enumerate(layers):  # no usable model_config: ReLU hidden, linear last
            if l[3] is None:
                l[3] = "linear" if i == len(layers) - 1 else "relu"
        opt = None
        if "optimizer_weights" in f:
            try:
                ow = f["optimizer_weights"]
                wnames = [_s(w) for w in np.atleast_1d(ow.attrs["weight_names"])]
                vals = {w: np.asarray(ow[w][()] if ow[w].shape == () else ow[w][:]) for w in wnames}
                it = [v for k, v in vals.items() if k.endswith("iter:0")]
                ms = [vals[w].ravel() for w in wnames if w.endswith("/m:0")]
                vs = [vals[w].ravel() for w in wnames if w.endswith("/v:0")]
                opt = {"iter": int(it[0]) if it else 0,
                       "m": np.concatenate(ms).astype(np.float32) if ms else None,
                       "v": np.concatenate(vs).astype(np.float32) if vs else None}
                tc = f.attrs.get("training_config")
                if tc is not None:
                    opt["config"] = json.loads(_s(tc)).get("optimizer_config", {}).get("config", {})
            except (H5Unsupported, KeyError, ValueError):
                opt = None
        kinds = None
        try:
            kv = f.attrs.get("v21_layer_kinds")  # written by Model.save: marks variational heads
            kinds = _s(kv).split(",") if kv is not None else None
        except H5Unsupported:
            kinds = None
        return {"layers": [tuple(l) for l in layers], "optimizer": opt, "config": cfg, "kinds": kinds}


def load_model(path):
    """``tf.keras.models.load_model`` for the files the reference ships: a Sequential with the
    stored kernels, biases and activations.  Accepts the engine's own .npz too, and falls
    back to ``<stem>.npz`` next to a missing ``<stem>.h5`` (the packaged conversions of
    the reference's AE-path files)."""
    from .engine import sequential_from_arrays
    if not os.path.exists(path) and path.endswith(".h5") and os.path.exists(path[:-3] + ".npz"):
        path = path[:-3] + ".npz"
    if not os.path.exists(path):
        raise IOError("No file or directory found at %s" % path)
    if path.endswith(".npz"):
        d = np.load(path, allow_pickle=False)
        n = int(d["n_layers"])
        Ws, bs = [d["W%d" % i] for i in range(n)], [d["b%d" % i] for i in range(n)]
        acts = [str(d["act%d" % i]) if ("act%d" % i) in d.files else ("linear" if i == n - 1 else "relu") for i in range(n)]
        return sequential_from_arrays(Ws, bs, acts)
    info = read_keras_h5(path)
    Ws = [l[1] for l in info["layers"]]
    bs = [l[2] for l in info["layers"]]
    acts = [l[3] for l in info["layers"]]
    m = sequential_from_arrays(Ws, bs, acts, name=(info["config"] or {}).get("config", {}).get("name") if isinstance((info["config"] or {}).get("config"), dict) else None)
    for layer, l in zip(m._layers, info["layers"]):
        layer.name = l[0]
    if info.get("kinds") and len(info["kinds"]) == len(m._layers):
        from .engine import GaussianLatent
        for i, kind in enumerate(info["kinds"]):
            if kind == "gaussian":  # (in, 2*latent) kernel of a variational head
                old = m._layers[i]
                g = GaussianLatent(old.kernel.shape[1] // 2, name=old.name)
                g.kernel, g.bias, g.input_dim = old.kernel, old.bias, old.input_dim
                m._layers[i] = g
    m._loaded_optimizer = info["optimizer"]
    o = info["optimizer"]
    if o is not None:  # as tf.keras.models.load_model does: the optimizer comes back with its state
        from . import optimizers
        c = o.get("config") or {}
        m.optimizer = optimizers.Adam(c.get("learning_rate", 1e-3), c.get("beta_1", 0.9), c.get("beta_2", 0.999),
                                      c.get("epsilon", 1e-7))
        m.optimizer.iterations = int(o.get("iter", 0))
        m._restore_state = (o.get("m"), o.get("v"))
    return m
