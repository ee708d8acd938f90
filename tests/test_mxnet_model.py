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


def test_policy_value_net_writes_the_reference_checkpoint_layout(tmp_path):
    """PolicyValueNet.save_model / get_policy_param (policy_value_net_mxnet.py:301-309): the file is the reference's
    pickle of (arg_params, aux_params) NDArray dicts -- what human_play_mxnet.py and train_mxnet.py's init_model
    unpickle and hand to set_params(*model_params) -- and it reads back into the same parameters.  (The class needs
    a GPU to be constructed; the checkpoint methods only need its parameter table.)"""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", 15, 15, 9, 2, 128, seed=4, style="bench")
    net = PolicyValueNet.__new__(PolicyValueNet)
    net._params = {k: v.copy() for k, v in prm.items()}
    arg, aux = net.get_policy_param()
    assert set(aux) == {k for k in prm if k.endswith(("_mean", "_var"))} and set(arg) | set(aux) == set(prm)
    path = str(tmp_path / "current_policy.model")
    net.save_model(path)
    a2, x2 = mm._Unpickler(open(path, "rb"), encoding="latin1").load()      # the pair the reference pickles
    assert set(a2) == set(arg) and set(x2) == set(aux)
    assert all(isinstance(v, mm.NDArray) for v in list(a2.values()) + list(x2.values()))
    back = weights.load_params(path)
    for k in prm:
        np.testing.assert_array_equal(back[k], prm[k])
    # the flat format stays readable too
    net.save_model(str(tmp_path / "flat.model"), fmt="flat")
    flat = weights.load_params(str(tmp_path / "flat.model"))
    np.testing.assert_array_equal(flat["convB2_weight"], prm["convB2_weight"])
