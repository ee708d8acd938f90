"""Numerics of the fp32 Winograd F(4x4,3x3) trunk (the default kernel, trunk15_wino3) under stress, on the GPU.

Winograd's error grows with the magnitude of the transformed operands, so the everyday parity tests
(tests/test_gpu_net.py: two synthetic initialisations) say little about other weight scales.  Here the full
10-block network (policy_value_net_mxnet.py:70-102) is evaluated against the float64 oracle with

  * trunk weights x4 and x0.25 (BatchNorm variances scaled along, so activations keep their scale and only the
    Winograd-domain operands grow / shrink), and raw x4 / x0.25 (activations explode / vanish through the layers),
  * BatchNorm moving variances of 1e-3 (1/sqrt(var + eps) = 22: every layer amplifies) and 10,
  * dense boards (every cell occupied) and empty boards,

and the direct-convolution kernel (APZ_TRUNK_KERNEL=ring: exact fp32 FMA chains, no transform) runs beside it.
Tolerance: north_star's 1e-4 on the logits, relative to the logit scale when that exceeds 1 (a network whose
logits are 1e6 cannot be held to 1e-4 absolute in fp32 by any kernel), with a 3x margin: the Winograd path must
stay below (1e-4 / 3) * max(1, max|logit|) (exploding networks: see the end of the test).  The measured table is
written to gpurun_out/ (committed as profiles/r03_winograd_numerics.json) and quoted in DESIGN.md.
"""
import json
import os

import numpy as np
import pytest

from alphapig_amd import weights
from oracle import net_ref
from test_gpu_net import _net_with_trunk_kernel

pytestmark = pytest.mark.gpu

TOL = 1e-4 / 3.0
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _boards(kind, n=6, seed=0):
    """9-plane inputs (game.py:68-94 layout is irrelevant here: the kernels see dense float planes)."""
    rs = np.random.RandomState(seed)
    planes = np.zeros((n, 9, 15, 15), np.float32)
    for i in range(n):
        if kind == "empty":
            planes[i, 8] = i % 2
            continue
        k = 225 if kind == "dense" else rs.randint(10, 120)
        cells = rs.permutation(225)[:k]
        for hist in range(4):                       # four history steps, own / opponent stones
            upto = max(0, k - hist)
            for j, c in enumerate(cells[:upto]):
                planes[i, 6 - 2 * hist + (j % 2), c // 15, c % 15] = 1
        planes[i, 8] = k % 2
    return planes


def _variant(name):
    prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=21, style="bench")
    trunk_w = [k for k in prm if k.startswith(("convA", "convB")) and k.endswith("_weight")]
    trunk_var = [k for k in prm if k.startswith(("bnA", "bnB")) and k.endswith("_moving_var")]
    all_var = [k for k in prm if k.endswith(("_moving_var", "_var"))]
    if name in ("w4_balanced", "w025_balanced", "w4_raw", "w025_raw"):
        f = 4.0 if name.startswith("w4") else 0.25
        for k in trunk_w:
            prm[k] = (prm[k] * f).astype(np.float32)
        if name.endswith("balanced"):                # BN divides it out again: activations keep their scale
            for k in trunk_var:
                prm[k] = (prm[k] * f * f).astype(np.float32)
            for k in prm:
                if k.startswith(("convA", "convB")) and k.endswith("_bias") or \
                        k.startswith(("bnA", "bnB")) and k.endswith("_moving_mean"):
                    prm[k] = (prm[k] * f).astype(np.float32)
    elif name == "var_1e-3":
        for k in all_var:
            prm[k] = np.full_like(prm[k], 1e-3)
    elif name == "var_10":
        for k in all_var:
            prm[k] = np.full_like(prm[k], 10.0)
    return prm


CASES = [("base", "random"), ("base", "dense"), ("base", "empty"), ("w4_balanced", "random"), ("w025_balanced", "random"),
         ("w4_raw", "random"), ("w025_raw", "random"), ("var_1e-3", "random"), ("var_10", "random"), ("var_10", "dense")]


def test_winograd_trunk_keeps_a_3x_margin_under_stress():
    rows, worst = [], 0.0
    for vname, bname in CASES:
        prm = _variant(vname)
        planes = _boards(bname)
        o_logits, _, o_vlog, _ = net_ref.forward(prm, planes, "resnet", 10, np.float64)
        scale = max(1.0, float(np.abs(o_logits).max()))
        vscale = max(1.0, float(np.abs(o_vlog).max()))
        errs = {}
        for kind in ("wino3", "ring"):
            net = _net_with_trunk_kernel(kind, prm, 10, 16)
            try:
                logits, _, vlog, _ = net.forward_with_logits(planes)
            finally:
                net.close()
            assert np.isfinite(logits).all() and np.isfinite(vlog).all(), (vname, bname, kind)
            errs[kind] = (float(np.abs(logits - o_logits).max()) / scale, float(np.abs(vlog - o_vlog[:, 0]).max()) / vscale)
        rows.append({"weights": vname, "boards": bname, "logit_scale": scale, "value_logit_scale": vscale,
                     "wino3_logit_err_rel": errs["wino3"][0], "wino3_value_err_rel": errs["wino3"][1],
                     "ring_logit_err_rel": errs["ring"][0], "ring_value_err_rel": errs["ring"][1]})
        worst = max(worst, errs["wino3"][0], errs["wino3"][1])
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r03_winograd_numerics.json"), "w") as f:
        json.dump({"tolerance": TOL, "rows": rows, "worst_wino3": worst}, f, indent=1)
    for r in rows:
        print("%-14s %-7s scale %9.3g  wino3 %.2e / %.2e   ring %.2e / %.2e" % (
            r["weights"], r["boards"], r["logit_scale"], r["wino3_logit_err_rel"], r["wino3_value_err_rel"],
            r["ring_logit_err_rel"], r["ring_value_err_rel"]))
    # Networks whose activations explode (raw x4 weights: logits 2e9; variances 1e-3: logits 2e26) sum terms that are
    # many orders larger than the result in the heads, so the relative error of EVERY fp32 kernel grows there, the
    # direct one's too (measured 1.4e-5 on the value logit); for those the Winograd path is held to 4x the direct
    # kernel's own error instead.
    def bound(r, key):
        exploding = r["logit_scale"] > 1e3
        return max(TOL, 4.0 * r["ring_" + key]) if exploding else TOL
    bad = [r for r in rows if r["wino3_logit_err_rel"] > bound(r, "logit_err_rel") or
           r["wino3_value_err_rel"] > bound(r, "value_err_rel")]
    assert not bad, bad
