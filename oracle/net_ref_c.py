"""ORACLE (test infrastructure): ctypes wrapper of oracle/net_ref.c -- the float32, one-position-
at-a-time CPU port of the residual net used as bench.py's `cpu_baseline` and cross-checked against
oracle/net_ref.py.  Built by alphapig_amd.build.build_oracle() into oracle/_build/."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libnet_ref.so")


def pack_params(prm, n_blocks):
    order = ["res_conv1_weight", "res_conv1_bias", "res_conv1_gamma", "res_conv1_beta", "res_conv1_mean",
             "res_conv1_var"]
    for i in range(1, n_blocks + 1):
        for ab in "AB":
            order += ["conv%s%d_weight" % (ab, i), "conv%s%d_bias" % (ab, i), "bn%s%d_gamma" % (ab, i),
                      "bn%s%d_beta" % (ab, i), "bn%s%d_moving_mean" % (ab, i), "bn%s%d_moving_var" % (ab, i)]
    for head, fc in (("conv3_1_1", "fc_3_1_1"), ("conv3_2_1", "fc_3_2_1")):
        order += [head + s for s in ("_weight", "_bias", "_gamma", "_beta", "_mean", "_var")]
        order += [fc + "_weight", fc + "_bias"]
    return np.concatenate([np.asarray(prm[k], dtype=np.float32).ravel() for k in order])


class CNet(object):
    def __init__(self, prm, height, width, c_in=9, n_filter=128, n_blocks=10, fast=False):
        if not os.path.exists(LIB):
            raise RuntimeError("%s missing: run `python -m alphapig_amd.build oracle`" % LIB)
        self.L = C.CDLL(LIB)
        fp = C.POINTER(C.c_float)
        self.L.ref_net_forward.restype = C.c_int
        self.L.ref_net_forward.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, fp]
        self.H, self.W, self.C, self.F, self.nb = height, width, c_in, n_filter, n_blocks
        self.blob = pack_params(prm, n_blocks)
        # vectorised forward (AVX-512 or AVX2+FMA, picked at run time); None -> plain loops
        self.L.ref_fast_create.restype = C.c_void_p
        self.L.ref_fast_create.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        self.L.ref_fast_forward.restype = C.c_int
        self.L.ref_fast_forward.argtypes = [C.c_void_p, fp, fp, fp, fp, fp]
        self.L.ref_fast_destroy.argtypes = [C.c_void_p]
        self.L.ref_fast_isa.argtypes = [C.c_void_p]
        self.fast = self.L.ref_fast_create(self.blob.ctypes.data_as(fp), c_in, n_filter, n_blocks, height, width) if fast else None
        self.isa = {0: "plain C loops", 1: "AVX2+FMA", 2: "AVX-512"}[self.L.ref_fast_isa(self.fast) if self.fast else 0]

    def __del__(self):
        if getattr(self, "fast", None):
            self.L.ref_fast_destroy(self.fast)
            self.fast = None

    def forward_one(self, planes):
        x = np.ascontiguousarray(planes, dtype=np.float32).reshape(self.C, self.H, self.W)
        hw = self.H * self.W
        probs = np.empty(hw, np.float32)
        value = np.empty(1, np.float32)
        logits = np.empty(hw, np.float32)
        vlogit = np.empty(1, np.float32)
        fp = C.POINTER(C.c_float)
        if self.fast:
            rc = self.L.ref_fast_forward(self.fast, x.ctypes.data_as(fp), probs.ctypes.data_as(fp), value.ctypes.data_as(fp),
                                         logits.ctypes.data_as(fp), vlogit.ctypes.data_as(fp))
        else:
            rc = self.L.ref_net_forward(self.blob.ctypes.data_as(fp), self.C, self.F, self.nb, self.H, self.W,
                                        x.ctypes.data_as(fp), probs.ctypes.data_as(fp), value.ctypes.data_as(fp),
                                        logits.ctypes.data_as(fp), vlogit.ctypes.data_as(fp))
        if rc != 0:
            raise MemoryError("ref_net_forward failed")
        return logits, probs, vlogit, value

    def policy_value_fn(self, board):
        """Same contract as the reference evaluator (policy_value_net_mxnet.py:261-280)."""
        legal = board.availables
        _, probs, _, value = self.forward_one(board.current_state())
        return zip(legal, probs[legal]), value
