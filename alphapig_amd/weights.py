"""Parameter tables and initialisers for the two policy-value networks.

Parameter names and shapes are the MXNet ones (reference policy_value_net_mxnet.py:41-102,
policy_value_loss.json; policy_value_net_mxnet_simple.py:39-92) so that a parameter dict
{name: float32 array} maps one-to-one onto the reference's (arg_params, aux_params).
"""
import collections
import pickle

import numpy as np

SIMPLE_LAYERS = (("conv1", 64), ("conv2", 64), ("conv3", 128), ("conv4", 128), ("conv5", 256),
                 ("conv_final", 256))


def param_shapes(kind, height, width, c_in=9, n_blocks=10, n_filter=128):
    """Ordered {name: shape}.  kind: 'resnet' | 'simple'."""
    hw = height * width
    sh = collections.OrderedDict()

    def conv_act(name, cin, cout, k):
        sh[name + "_weight"] = (cout, cin, k, k)
        sh[name + "_bias"] = (cout,)
        for s in ("_gamma", "_beta", "_mean", "_var"):
            sh[name + s] = (cout,)

    if kind == "resnet":
        conv_act("res_conv1", c_in, n_filter, 3)
        for i in range(1, n_blocks + 1):
            for ab in "AB":
                sh["conv%s%d_weight" % (ab, i)] = (n_filter, n_filter, 3, 3)
                sh["conv%s%d_bias" % (ab, i)] = (n_filter,)
                for s in ("_gamma", "_beta", "_moving_mean", "_moving_var"):
                    sh["bn%s%d%s" % (ab, i, s)] = (n_filter,)
        last = n_filter
    elif kind == "simple":
        prev = c_in
        for name, co in SIMPLE_LAYERS:
            conv_act(name, prev, co, 3)
            prev = co
        last = prev
    else:
        raise ValueError("kind must be 'resnet' or 'simple'")
    conv_act("conv3_1_1", last, 4, 1)
    sh["fc_3_1_1_weight"] = (hw, 4 * hw)
    sh["fc_3_1_1_bias"] = (hw,)
    conv_act("conv3_2_1", last, 2, 1)
    sh["fc_3_2_1_weight"] = (1, 2 * hw)
    sh["fc_3_2_1_bias"] = (1,)
    return sh


def _xavier(rs, shape):
    """mx.init.Xavier() defaults (rnd_type uniform, factor_type avg, magnitude 3), the
    initialiser the reference uses (policy_value_net_mxnet.py:205)."""
    hw_scale = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[1] * hw_scale, shape[0] * hw_scale
    bound = np.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
    return rs.uniform(-bound, bound, size=shape).astype(np.float32)


def init_params(kind, height, width, c_in=9, n_blocks=10, n_filter=128, seed=0, style="reference"):
    """style 'reference': what a freshly constructed reference net holds (Xavier weights, zero
    biases / beta / means, unit gamma / var).  style 'bench': SURVEY.md 8(d) non-trivial
    statistics (biases N(0,.05), gamma U(.5,1.5), beta N(0,.1), mean N(0,.1), var U(.5,1.5))
    so that BatchNorm folding mistakes show up in parity tests."""
    rs = np.random.RandomState(seed)
    out = collections.OrderedDict()
    for name, shape in param_shapes(kind, height, width, c_in, n_blocks, n_filter).items():
        if name.endswith("_weight"):
            out[name] = _xavier(rs, shape)
        elif style == "reference":
            one = name.endswith("_gamma") or name.endswith("_var")
            out[name] = (np.ones if one else np.zeros)(shape, dtype=np.float32)
        elif name.endswith("_bias"):
            out[name] = rs.normal(0, 0.05, size=shape).astype(np.float32)
        elif name.endswith("_gamma") or name.endswith("_var"):
            out[name] = rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
        else:   # beta, mean
            out[name] = rs.normal(0, 0.1, size=shape).astype(np.float32)
    return out


def save_params(params, path):
    """Pickle protocol 2 of {name: ndarray} (the reference pickles MXNet NDArrays the same
    way, policy_value_net_mxnet.py:305-309; an MXNet-NDArray reader is a 'next' row)."""
    with open(path, "wb") as f:
        pickle.dump({k: np.asarray(v, dtype=np.float32) for k, v in params.items()}, f, protocol=2)


def load_params(path):
    """Own pickles ({name: ndarray}) or the reference's MXNet `.model` files (dicts of
    mx.nd.NDArray, read without MXNet by alphapig_amd.mxnet_model)."""
    try:
        with open(path, "rb") as f:
            obj = pickle.load(f)
    except (ImportError, ModuleNotFoundError, AttributeError):
        from . import mxnet_model
        return mxnet_model.load_model(path)
    if isinstance(obj, (tuple, list)) and len(obj) == 2:      # (arg_params, aux_params)
        merged = dict(obj[0])
        merged.update(obj[1])
        obj = merged
    return collections.OrderedDict((k, np.asarray(v, dtype=np.float32)) for k, v in obj.items())
