"""Pure-Python writer of the HDF5 subset that Keras legacy-H5 model files use: the mirror of
``h5lite`` (the reader), so that ``Model.save(path)`` produces a file
``tf.keras.models.load_model`` / h5py / ``h5lite.load_model`` can read -- kernels, biases,
layer names, activations and the Adam state (iter, m, v).

Why it exists: the reference loads and ships such files (emulator.py:335-337, :691-693;
models/autoencoder_based_emulator/*.h5) but there is no h5py in the product environment;
SURVEY section 8(f) ranks the reader/writer second among the pieces around the hot path.

What is written (HDF5 "earliest" feature set, the one h5py's defaults read everywhere):
superblock version 0; old-style groups (symbol-table message -> one B-tree node -> one
symbol-table node -> local heap); version-1 object headers; contiguous little-endian datasets
(float32, int64); version-1 attributes holding fixed-length strings (scalar or 1-D).
No chunking, compression, variable-length data or free-space management.
"""
import json
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K = 32      # up to 64 links per group in a single symbol-table node
INTERNAL_K = 16


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


# ---- datatype / dataspace messages -------------------------------------------------------
def _dtype_msg(dt, strlen=None):
    if strlen is not None:  # fixed-length, null-padded ASCII string
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, strlen)
    dt = np.dtype(dt)
    if dt == np.float32:
        return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    if dt == np.float64:
        return struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    if dt == np.int64:
        return struct.pack("<BBBBI", 0x10, 0x08, 0, 0, 8) + struct.pack("<HH", 0, 64)
    if dt == np.int32:
        return struct.pack("<BBBBI", 0x10, 0x08, 0, 0, 4) + struct.pack("<HH", 0, 32)
    raise TypeError("h5write: dtype %r not supported" % (dt,))


def _space_msg(shape):
    shape = tuple(int(s) for s in shape)
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", s) for s in shape)


def _attr_msg(name, value):
    """value: str/bytes (scalar string), list of str (1-D strings), or a numpy scalar/array."""
    nm = name.encode() + b"\0"
    if isinstance(value, (str, bytes)):
        raw = value.encode() if isinstance(value, str) else value
        n = max(1, len(raw))
        dtm, spm, data = _dtype_msg(None, n), _space_msg(()), raw.ljust(n, b"\0")
    elif isinstance(value, (list, tuple)) and all(isinstance(v, (str, bytes)) for v in value):
        raws = [v.encode() if isinstance(v, str) else v for v in value]
        n = max([1] + [len(r) for r in raws])
        dtm, spm = _dtype_msg(None, n), _space_msg((len(raws),))
        data = b"".join(r.ljust(n, b"\0") for r in raws)
    else:
        arr = np.require(np.asarray(value), requirements="C")
        dtm, spm, data = _dtype_msg(arr.dtype), _space_msg(arr.shape), arr.astype(arr.dtype.newbyteorder("<")).tobytes()
    body = struct.pack("<BBHHH", 1, 0, len(nm), len(dtm), len(spm)) + _pad8(nm) + _pad8(dtm) + _pad8(spm) + data
    if len(body) > 0xFFF0:
        raise ValueError("h5write: attribute %r is %d bytes; object-header messages hold < 64 KiB" % (name, len(body)))
    return 0x000C, body


def _header(msgs):
    """Version-1 object header from [(type, body)]."""
    blob = b""
    for typ, body in msgs:
        body = _pad8(body)
        blob += struct.pack("<HHB3x", typ, len(body), 0) + body
    return struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(blob)) + blob


class _Node:
    def __init__(self):
        self.attrs = {}


class GroupW(_Node):
    def __init__(self):
        super().__init__()
        self.children = {}

    def create_group(self, name):
        g = self
        for part in name.strip("/").split("/"):
            g = g.children.setdefault(part, GroupW())
            if not isinstance(g, GroupW):
                raise ValueError("%r is a dataset" % part)
        return g

    def create_dataset(self, name, data):
        parts = name.strip("/").split("/")
        g = self.create_group("/".join(parts[:-1])) if len(parts) > 1 else self
        d = DatasetW(np.asarray(data))
        g.children[parts[-1]] = d
        return d


class DatasetW(_Node):
    def __init__(self, data):
        super().__init__()
        if data.dtype not in (np.float32, np.float64, np.int64, np.int32):
            raise TypeError("h5write: dataset dtype %r not supported" % (data.dtype,))
        self.data = np.require(data, requirements="C")  # (ascontiguousarray would turn a scalar into shape (1,))


class FileW(GroupW):
    """Build the tree in memory (create_group / create_dataset / .attrs), then ``write(path)``."""

    def write(self, path):
        buf = bytearray(96)  # superblock, patched at the end

        def alloc(b):
            while len(buf) % 8:
                buf.append(0)
            off = len(buf)
            buf.extend(b)
            return off

        def emit(node):
            """-> (object header address, symbol-table scratch or None)"""
            amsgs = [_attr_msg(k, v) for k, v in node.attrs.items()]
            if isinstance(node, DatasetW):
                raw = node.data.astype(node.data.dtype.newbyteorder("<")).tobytes()
                daddr = alloc(raw) if raw else UNDEF
                msgs = [(0x0001, _space_msg(node.data.shape)), (0x0003, _dtype_msg(node.data.dtype)),
                        (0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),                      # fill value v2: none defined
                        (0x0008, struct.pack("<BBQQ", 3, 1, daddr, len(raw)))] + amsgs  # contiguous layout v3
                return alloc(_header(msgs)), None
            names = sorted(node.children, key=lambda s: s.encode())
            if len(names) > 2 * LEAF_K:
                raise ValueError("h5write: a group holds at most %d links" % (2 * LEAF_K))
            kids = [(n,) + emit(node.children[n]) for n in names]
            # local heap: "" at offset 0, then the link names; one free block closes the segment
            heap, offs = bytearray(b"\0" * 8), {}
            for n in names:
                offs[n] = len(heap)
                heap.extend(_pad8(n.encode() + b"\0"))
            free_off = len(heap)
            heap.extend(struct.pack("<QQ", 1, 16))
            heap_data = alloc(bytes(heap))
            heap_addr = alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data))
            snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(kids))
            for n, addr, scratch in kids:
                if scratch is None:
                    snod += struct.pack("<QQII16x", offs[n], addr, 0, 0)
                else:
                    snod += struct.pack("<QQIIQQ", offs[n], addr, 1, 0, scratch[0], scratch[1])
            snod = snod.ljust(8 + 2 * LEAF_K * 40, b"\0")
            snod_addr = alloc(snod)
            tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if kids else 0, UNDEF, UNDEF)
            tree += struct.pack("<QQQ", 0, snod_addr, offs[names[-1]] if names else 0)
            tree = tree.ljust(24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8, b"\0")
            tree_addr = alloc(tree)
            hdr = alloc(_header([(0x0011, struct.pack("<QQ", tree_addr, heap_addr))] + amsgs))
            return hdr, (tree_addr, heap_addr)

        root_hdr, root_scratch = emit(self)
        while len(buf) % 8:
            buf.append(0)
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(buf), UNDEF)
        sb += struct.pack("<QQIIQQ", 0, root_hdr, 1, 0, root_scratch[0], root_scratch[1])
        assert len(sb) == 96
        buf[:96] = sb
        with open(path, "wb") as f:
            f.write(bytes(buf))


# ---- Keras legacy-H5 model files ------------------------------------------------------
def _keras_model_config(names, kernels, acts, model_name):
    layers = [{"class_name": "InputLayer",
               "config": {"batch_input_shape": [None, int(kernels[0].shape[0])], "dtype": "float32", "sparse": False,
                          "ragged": False, "name": "input_1"}}]
    for n, k, a in zip(names, kernels, acts):
        layers.append({"class_name": "Dense",
                       "config": {"name": n, "trainable": True, "dtype": "float32", "units": int(k.shape[1]),
                                  "activation": a, "use_bias": True,
                                  "kernel_initializer": {"class_name": "GlorotUniform", "config": {"seed": None}},
                                  "bias_initializer": {"class_name": "Zeros", "config": {}},
                                  "kernel_regularizer": None, "bias_regularizer": None, "activity_regularizer": None,
                                  "kernel_constraint": None, "bias_constraint": None}})
    return {"class_name": "Sequential", "config": {"name": model_name or "sequential", "layers": layers}}


def write_keras_h5(path, names, kernels, biases, acts, model_name=None, optimizer=None, extra_attrs=None):
    """The layout ``h5lite.read_keras_h5`` documents.  ``optimizer``: None or
    dict(iter=int, m=flat float32, v=flat float32, config={learning_rate, beta_1, beta_2, epsilon}) in arena
    order (kernel, bias per layer)."""
    f = FileW()
    f.attrs["keras_version"] = "2.7.0"
    f.attrs["backend"] = "tensorflow"
    f.attrs["model_config"] = json.dumps(_keras_model_config(names, kernels, acts, model_name))
    for k, v in (extra_attrs or {}).items():
        f.attrs[k] = v
    mw = f.create_group("model_weights")
    mw.attrs["layer_names"] = list(names)
    mw.attrs["backend"] = "tensorflow"
    mw.attrs["keras_version"] = "2.7.0"
    for n, k, b in zip(names, kernels, biases):
        g = mw.create_group(n)
        g.attrs["weight_names"] = ["%s/kernel:0" % n, "%s/bias:0" % n]
        g.create_dataset("%s/kernel:0" % n, np.asarray(k, np.float32))
        g.create_dataset("%s/bias:0" % n, np.asarray(b, np.float32))
    if optimizer is not None:
        cfg = dict(optimizer.get("config") or {})
        tc = {"loss": None, "metrics": None, "weighted_metrics": None, "loss_weights": None,
              "optimizer_config": {"class_name": "Adam",
                                   "config": {"name": "Adam", "learning_rate": float(cfg.get("learning_rate", 1e-3)),
                                              "decay": 0.0, "beta_1": float(cfg.get("beta_1", 0.9)),
                                              "beta_2": float(cfg.get("beta_2", 0.999)),
                                              "epsilon": float(cfg.get("epsilon", 1e-7)), "amsgrad": False}}}
        f.attrs["training_config"] = json.dumps(tc)
        ow = f.create_group("optimizer_weights")
        wn = ["Adam/iter:0"]
        ow.create_dataset("Adam/iter:0", np.array(int(optimizer.get("iter", 0)), np.int64))
        m, v = np.asarray(optimizer["m"], np.float32), np.asarray(optimizer["v"], np.float32)
        for slot, flat in (("m", m), ("v", v)):  # Keras order: every m (kernel, bias per layer), then every v
            o = 0
            for n, k, b in zip(names, kernels, biases):
                for part, arr in (("kernel", k), ("bias", b)):
                    name = "Adam/%s/%s/%s:0" % (n, part, slot)
                    ow.create_dataset(name, flat[o:o + arr.size].reshape(arr.shape))
                    wn.append(name)
                    o += arr.size
        ow.attrs["weight_names"] = wn
    f.write(path)
