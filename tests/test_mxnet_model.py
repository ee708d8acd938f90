"""MXNet-free `.model` checkpoint reader/writer.  UNVERIFIED against real MXNet
files (none exist, MXNet is not installable here): byte layout restated from the MXNet 1.x
sources, checked by hand-built blobs and round trips."""
import pickle
import struct

import numpy as np
import pytest

from alphapig_amd import mxnet_model as mm
from alphapig_amd import weights


def test_ndarray_blob_layouts():
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4) * 0.5
    np.testing.assert_array_equal(mm.decode_ndarray(mm.encode_ndarray(a)), a)
    # V2 blob spelled out by hand: magic, stype, ndim, int64 dims, ctx(dev_type, dev_id), type flag, data
    raw = struct.pack("<IiiqqiiI", 0xF993FAC9, 0, 2, 2, 3, 2, 0, 0)[:-4] + struct.pack("<i", 0) + \
        np.arange(6, dtype="<f4").tobytes()
    np.testing.assert_array_equal(mm.decode_ndarray(raw), np.arange(6, dtype=np.float32).reshape(2, 3))
    # V1 (legacy) blob: uint32 dims, no storage type
    raw1 = struct.pack("<IIII", 0xF993FAC8, 2, 3, 2) + struct.pack("<iii", 1, 0, 1) + np.arange(6, dtype="<f8").tobytes()
    got = mm.decode_ndarray(raw1)
    assert got.dtype == np.float64 and got.shape == (3, 2)
    with pytest.raises(mm.MXNetFormatError):
        mm.decode_ndarray(b"\x00" * 40)                      # bad magic
    with pytest.raises(mm.MXNetFormatError):
        mm.decode_ndarray(mm.encode_ndarray(a)[:-4])          # truncated payload
    bad = bytearray(mm.encode_ndarray(a))
    struct.pack_into("<i", bad, 4, 1)                        # sparse storage type
    with pytest.raises(mm.MXNetFormatError):
        mm.decode_ndarray(bytes(bad))


def test_model_file_round_trip(tmp_path):
    prm = weights.init_params("resnet", 8, 8, 9, 1, 64, seed=2, style="bench")
    path = str(tmp_path / "best_policy_7.model")
    mm.save_model(prm, path)
    blob = open(path, "rb").read()
    assert b"mxnet.ndarray.ndarray" in blob and b"NDArray" in blob and blob[:2] == b"\x80\x02"   # protocol 2
    with pytest.raises((ImportError, ModuleNotFoundError, AttributeError)):
        pickle.loads(blob)                                   # a plain unpickler needs MXNet
    back = mm.load_model(path)
    assert set(back) == set(prm)
    for k in prm:
        np.testing.assert_array_equal(back[k], prm[k])
    # split follows Module.get_params(): moving statistics are aux params
    arg, aux = mm._Unpickler(open(path, "rb"), encoding="latin1").load()
    assert "bnA1_moving_mean" in aux and "res_conv1_mean" in aux and "convA1_weight" in arg
    # the generic loader falls through to the MXNet reader
    again = weights.load_params(path)
    np.testing.assert_array_equal(again["fc_3_1_1_weight"], prm["fc_3_1_1_weight"])
    import sys
    assert "mxnet" not in sys.modules
