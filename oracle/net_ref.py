"""ORACLE (test infrastructure): NumPy restatement of the policy-value networks.

PARITY UNPINNED: the reference's arithmetic is MXNet 1.6.0 (requirements.txt:8; graph
JSON written by 1.5.1), absent from /root/reference; the reference ships no weights and
no recorded activations.  This file restates

  * the residual net   /root/reference/policy_value_net_mxnet.py:41-102
    (node order / attrs / parameter names cross-checked with policy_value_loss.json:
     BatchNorm nodes 9, 208, 226 carry no attrs -> MXNet default fix_gamma=True;
     bnA*/bnB* carry fix_gamma=False), and
  * the plain 6-conv net /root/reference/policy_value_net_mxnet_simple.py:68-92

with the published MXNet-1.x operator semantics: Convolution = NCHW cross-correlation,
stride 1, zero pad k//2, bias added; BatchNorm(inference) = (x-moving_mean) /
sqrt(moving_var + 1e-3) * gamma + beta with gamma := 1 when fix_gamma; FullyConnected
y = flatten(x) @ W.T + b (row-major flatten of [C,H,W]); SoftmaxActivation over the
row; Dropout = identity at inference; tanh on the value.

BN is applied UNFOLDED here (the HIP path folds it into the conv weights), in the
dtype given (float64 = de-facto oracle, float32 = what a CPU port would compute).
"""
import numpy as np

BN_EPS = 1e-3

SIMPLE_LAYERS = (("conv1", 64), ("conv2", 64), ("conv3", 128), ("conv4", 128),
                 ("conv5", 256), ("conv_final", 256))


def _conv(x, w, b):
    """x [N,Ci,H,W], w [Co,Ci,k,k], b [Co] -> [N,Co,H,W] (cross-correlation, same pad)."""
    n, ci, h, wd = x.shape
    co, _, k, _ = w.shape
    p = k // 2
    xp = np.zeros((n, ci, h + 2 * p, wd + 2 * p), dtype=x.dtype)
    xp[:, :, p:p + h, p:p + wd] = x
    out = np.zeros((n, co, h, wd), dtype=x.dtype)
    for ky in range(k):
        for kx in range(k):
            patch = xp[:, :, ky:ky + h, kx:kx + wd]                      # [N,Ci,H,W]
            out += np.einsum("nchw,oc->nohw", patch, w[:, :, ky, kx], optimize=True)
    return out + b[None, :, None, None]


def _bn(x, prm, name, fix_gamma, stat_names):
    mean_n, var_n = stat_names
    dt = x.dtype
    gamma = np.ones_like(prm[name + "_beta"], dtype=dt) if fix_gamma else prm[name + "_gamma"].astype(dt)
    beta = prm[name + "_beta"].astype(dt)
    mean = prm[name + mean_n].astype(dt)
    var = prm[name + var_n].astype(dt)
    inv = (1.0 / np.sqrt(var + dt.type(BN_EPS))).astype(dt)
    sh = (1, -1, 1, 1)
    return (x - mean.reshape(sh)) * inv.reshape(sh) * gamma.reshape(sh) + beta.reshape(sh)


def _conv_act(x, prm, name):
    """conv_act() of both reference files (:41-60): conv + BN(fix_gamma default) + relu."""
    dt = x.dtype
    y = _conv(x, prm[name + "_weight"].astype(dt), prm[name + "_bias"].astype(dt))
    y = _bn(y, prm, name, True, ("_mean", "_var"))
    return np.maximum(y, 0)


def _heads(x, prm):
    dt = x.dtype
    n = x.shape[0]
    pol = _conv_act(x, prm, "conv3_1_1").reshape(n, -1)
    logits = pol @ prm["fc_3_1_1_weight"].astype(dt).T + prm["fc_3_1_1_bias"].astype(dt)
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    probs = e / e.sum(axis=1, keepdims=True)
    val = _conv_act(x, prm, "conv3_2_1").reshape(n, -1)
    vlogit = val @ prm["fc_3_2_1_weight"].astype(dt).T + prm["fc_3_2_1_bias"].astype(dt)
    return logits, probs, vlogit, np.tanh(vlogit)


def forward_resnet(prm, planes, n_blocks, dtype=np.float64, return_trunk=False):
    """policy_value_net_mxnet.py:70-102."""
    x = np.asarray(planes).astype(dtype)
    x = _conv_act(x, prm, "res_conv1")
    stem = x
    for i in range(1, n_blocks + 1):
        skip = x
        y = _conv(x, prm["convA%d_weight" % i].astype(dtype), prm["convA%d_bias" % i].astype(dtype))
        y = np.maximum(_bn(y, prm, "bnA%d" % i, False, ("_moving_mean", "_moving_var")), 0)
        y = _conv(y, prm["convB%d_weight" % i].astype(dtype), prm["convB%d_bias" % i].astype(dtype))
        y = _bn(y, prm, "bnB%d" % i, False, ("_moving_mean", "_moving_var"))
        x = np.maximum(y + skip, 0)
    out = _heads(x, prm)
    return out + ((stem, x),) if return_trunk else out


def forward_simple(prm, planes, dtype=np.float64, return_trunk=False):
    """policy_value_net_mxnet_simple.py:68-92."""
    x = np.asarray(planes).astype(dtype)
    first = None
    for name, _ in SIMPLE_LAYERS:
        x = _conv_act(x, prm, name)
        if first is None:
            first = x
    out = _heads(x, prm)
    return out + ((first, x),) if return_trunk else out


def forward(prm, planes, kind="resnet", n_blocks=10, dtype=np.float64, return_trunk=False):
    """-> (logits [N,HW], probs [N,HW], value_logit [N,1], value [N,1]) [+ (stem, trunk)]."""
    if kind == "resnet":
        return forward_resnet(prm, planes, n_blocks, dtype, return_trunk)
    return forward_simple(prm, planes, dtype, return_trunk)


def flops_per_leaf(kind, h, w, c_in=9, n_blocks=10, n_filter=128):
    """Multiply-accumulate count *2."""
    hw = h * w
    if kind == "resnet":
        mac = c_in * n_filter * 9 * hw + n_blocks * 2 * n_filter * n_filter * 9 * hw
        last = n_filter
    else:
        mac, prev = 0, c_in
        for _, co in SIMPLE_LAYERS:
            mac += prev * co * 9 * hw
            prev = co
        last = prev
    mac += last * 6 * hw + 4 * hw * hw + 2 * hw
    return 2 * mac
