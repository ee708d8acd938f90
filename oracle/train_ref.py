"""ORACLE (test infrastructure): NumPy float64 restatement of the TRAINING graph's loss and of
the MXNet Adam update -- PARITY UNPINNED (MXNet absent; see oracle/__init__.py).

loss follows policy_value_net_mxnet.py:173-194 on the training-mode graph :70-102 (BatchNorm with
batch statistics, biased variance, eps 1e-3; fix_gamma as in oracle/net_ref.py; dropout left out:
the tests run the trainer with dropout 0 when comparing).  Gradients come from central finite
differences of this loss -- an independent check of the autograd graph in alphapig_amd/train.py.
"""
import numpy as np

from oracle.net_ref import BN_EPS, _conv


def _bn_train(x, prm, name, fix_gamma):
    mean = x.mean(axis=(0, 2, 3), keepdims=True)
    var = x.var(axis=(0, 2, 3), keepdims=True)
    gamma = 1.0 if fix_gamma else prm[name + "_gamma"].reshape(1, -1, 1, 1)
    return (x - mean) / np.sqrt(var + BN_EPS) * gamma + prm[name + "_beta"].reshape(1, -1, 1, 1)


def _conv_act(x, prm, name):
    y = _conv(x, prm[name + "_weight"].astype(np.float64), prm[name + "_bias"].astype(np.float64))
    return np.maximum(_bn_train(y, prm, name, True), 0)


def train_loss(prm, states, pis, zs, n_blocks):
    """-> (loss, entropy) of the residual net in training mode (no dropout), float64."""
    prm = {k: np.asarray(v, dtype=np.float64) for k, v in prm.items()}
    x = _conv_act(np.asarray(states, dtype=np.float64), prm, "res_conv1")
    for i in range(1, n_blocks + 1):
        skip = x
        y = _conv(x, prm["convA%d_weight" % i], prm["convA%d_bias" % i])
        y = np.maximum(_bn_train(y, prm, "bnA%d" % i, False), 0)
        y = _conv(y, prm["convB%d_weight" % i], prm["convB%d_bias" % i])
        x = np.maximum(_bn_train(y, prm, "bnB%d" % i, False) + skip, 0)
    n = x.shape[0]
    pol = _conv_act(x, prm, "conv3_1_1").reshape(n, -1)
    val = _conv_act(x, prm, "conv3_2_1").reshape(n, -1)
    logits = pol @ prm["fc_3_1_1_weight"].T + prm["fc_3_1_1_bias"]
    logits = logits - logits.max(axis=1, keepdims=True)
    logp = logits - np.log(np.exp(logits).sum(axis=1, keepdims=True))
    v = np.tanh(val @ prm["fc_3_2_1_weight"].T + prm["fc_3_2_1_bias"])
    value_loss = np.mean((np.asarray(zs, dtype=np.float64).reshape(-1, 1) - v) ** 2)
    policy_loss = np.mean(-np.sum(logp * np.asarray(pis, dtype=np.float64), axis=1))
    entropy = np.mean(np.sum(-np.exp(logp) * logp, axis=1))
    return value_loss + policy_loss, entropy


def finite_difference(prm, name, index, states, pis, zs, n_blocks, h=1e-5):
    p = {k: np.array(v, dtype=np.float64) for k, v in prm.items()}
    base = p[name].flat[index]
    p[name].flat[index] = base + h
    up = train_loss(p, states, pis, zs, n_blocks)[0]
    p[name].flat[index] = base - h
    dn = train_loss(p, states, pis, zs, n_blocks)[0]
    return (up - dn) / (2 * h)


def adam_step(w, g, m, v, t, lr, wd, batch_size, b1=0.9, b2=0.999, eps=1e-8):
    """MXNet Adam as the reference configures it (rescale_grad = 1/batch_size, wd folded into g)."""
    g = g / batch_size + wd * w
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return w - lr_t * m / (np.sqrt(v) + eps), m, v
