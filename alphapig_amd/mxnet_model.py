"""Reader / writer for the reference's `.model` checkpoints WITHOUT MXNet.

The reference saves `pickle.dump((arg_params, aux_params), f, protocol=2)` where both are dicts
{name: mx.nd.NDArray} (policy_value_net_mxnet.py:305-309) and loads them with
`pickle.load(open(model_file, 'rb'))` into `PolicyValueNet(..., model_params=...)`
(train_mxnet.py:313-316).  An NDArray pickles as `NDArray.__reduce__ -> (NDArray, (None,),
{'handle': bytearray})` whose bytes are MXNDArraySaveRawBytes, i.e. NDArray::Save.

UNVERIFIED FORMAT: no `.model` file exists in the reference tree and MXNet cannot be installed
here, so the byte layout below is restated from the MXNet 1.x sources (src/ndarray/ndarray.cc,
nnvm Tuple::Save, Context::Save) and checked only by round trip (tests/test_mxnet_model.py):

    uint32 magic            0xF993FAC9 (V2) / 0xF993FACA (V3, numpy shape semantics)
    int32  storage type     0 = dense (sparse arrays are rejected)
    int32  ndim, int64 dims[ndim]          (V1 0xF993FAC8: no storage type, uint32 ndim + uint32 dims)
    int32  dev_type, int32 dev_id          (context the array lived on)
    int32  type flag        0 f32, 1 f64, 2 f16, 3 u8, 4 i32, 5 i8, 6 i64
    raw little-endian data
The reader checks the magic and that the payload length matches exactly, so a layout it does
not understand fails loudly instead of producing wrong weights.
"""
import collections
import pickle
import struct

import numpy as np

V1, V2, V3 = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
_DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _DTYPES.items()}


class MXNetFormatError(ValueError):
    pass


def decode_ndarray(raw):
    """bytes of NDArray::Save -> numpy array."""
    raw = bytes(raw)
    if len(raw) < 8:
        raise MXNetFormatError("NDArray blob too short")
    magic, = struct.unpack_from("<I", raw, 0)
    off = 4
    if magic in (V2, V3):
        stype, = struct.unpack_from("<i", raw, off)
        off += 4
        if stype != 0:
            raise MXNetFormatError("sparse NDArray (storage type %d) is not supported" % stype)
        ndim, = struct.unpack_from("<i", raw, off)
        off += 4
        if ndim < 0 or ndim > 32:
            raise MXNetFormatError("implausible ndim %d" % ndim)
        shape = struct.unpack_from("<%dq" % ndim, raw, off)
        off += 8 * ndim
    elif magic == V1:
        ndim, = struct.unpack_from("<I", raw, off)
        off += 4
        if ndim > 32:
            raise MXNetFormatError("implausible ndim %d" % ndim)
        shape = struct.unpack_from("<%dI" % ndim, raw, off)
        off += 4 * ndim
    else:
        raise MXNetFormatError("unknown NDArray magic 0x%08X" % magic)
    off += 8                                             # context: dev_type, dev_id
    flag, = struct.unpack_from("<i", raw, off)
    off += 4
    if flag not in _DTYPES:
        raise MXNetFormatError("unknown type flag %d" % flag)
    dt = np.dtype(_DTYPES[flag]).newbyteorder("<")
    count = int(np.prod(shape, dtype=np.int64)) if ndim else 1
    if len(raw) - off != count * dt.itemsize:
        raise MXNetFormatError("payload is %d bytes, shape %s x %s needs %d" %
                               (len(raw) - off, tuple(shape), dt, count * dt.itemsize))
    return np.frombuffer(raw, dtype=dt, count=count, offset=off).reshape(shape).copy()


def encode_ndarray(a):
    a = np.ascontiguousarray(a)
    if a.dtype not in _FLAGS:
        a = a.astype(np.float32)
    head = struct.pack("<Ii", V2, 0) + struct.pack("<i", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape)
    head += struct.pack("<ii", 1, 0)                      # Context: cpu(0)
    head += struct.pack("<i", _FLAGS[a.dtype])
    return head + a.astype(a.dtype.newbyteorder("<")).tobytes()


class NDArray(object):
    """Stand-in for mxnet.ndarray.NDArray during (un)pickling."""

    def __init__(self, handle=None):
        self.array = None

    def __setstate__(self, state):
        h = state.get("handle") if isinstance(state, dict) else None
        self.array = None if h is None else decode_ndarray(h)

    def __reduce__(self):
        return NDArray, (None,), {"handle": bytearray(encode_ndarray(self.array))}

    def asnumpy(self):
        return self.array


NDArray.__module__ = "mxnet.ndarray.ndarray"                 # what a real MXNet unpickler will look up
NDArray.__qualname__ = "NDArray"


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name == "NDArray" and module.startswith("mxnet"):
            return NDArray
        return super().find_class(module, name)


def load_model(path_or_file):
    """-> OrderedDict {name: float32 ndarray} merged from (arg_params, aux_params)."""
    f = open(path_or_file, "rb") if isinstance(path_or_file, str) else path_or_file
    try:
        obj = _Unpickler(f, encoding="latin1").load()
    finally:
        if isinstance(path_or_file, str):
            f.close()
    parts = obj if isinstance(obj, (tuple, list)) else (obj,)
    out = collections.OrderedDict()
    for d in parts:
        for k, v in d.items():
            arr = v.array if isinstance(v, NDArray) else np.asarray(v)
            out[str(k)] = np.asarray(arr, dtype=np.float32)
    return out


AUX_SUFFIXES = ("_mean", "_var", "_moving_mean", "_moving_var")   # BatchNorm statistics: MXNet auxiliary states


def save_model(params, path, aux_suffixes=AUX_SUFFIXES):
    """Write {name: array} as the reference's (arg_params, aux_params) pickle of NDArrays
    (protocol 2).  BN moving statistics go to aux_params like MXNet's Module.get_params().
    pickle only writes classes it can import, so inert `mxnet.ndarray.ndarray` module objects
    holding the stand-in class are registered for the duration of the dump."""
    import sys
    import types
    arg, aux = {}, {}
    for k, v in params.items():
        nd = NDArray()
        nd.array = np.asarray(v, dtype=np.float32)
        (aux if k.endswith(aux_suffixes) else arg)[k] = nd
    names = ("mxnet", "mxnet.ndarray", "mxnet.ndarray.ndarray")
    saved = {n: sys.modules.get(n) for n in names}
    try:
        for n in names:
            if saved[n] is None:
                sys.modules[n] = types.ModuleType(n)
        if not hasattr(sys.modules["mxnet.ndarray.ndarray"], "NDArray"):
            sys.modules["mxnet.ndarray.ndarray"].NDArray = NDArray
        with open(path, "wb") as f:
            pickle.dump((arg, aux), f, protocol=2)
    finally:
        for n in names:
            if saved[n] is None:
                sys.modules.pop(n, None)
