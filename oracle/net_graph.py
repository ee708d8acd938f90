"""ORACLE (test infrastructure): the reference's network as a symbol graph, restated without MXNet.

The only machine-readable pin of the reference architecture is the graph it saves while
building its training module, /root/reference/policy_value_loss.json (written by
policy_value_net_mxnet.py:194, MXNet 1.5.1).  This module restates the symbol-building
code of policy_value_net_mxnet.py -- conv_act :41-60, fc_self :62-68,
create_backbone_resnet :70-102, create_policy_value_train :173-195 -- on a minimal
re-implementation of what MXNet's symbol composition does when it serialises a graph:

  * an operator that is given no symbol for one of its parameter inputs creates a variable
    `<name>_<arg>` that inherits the operator's attributes (nnvm Symbol::Compose);
  * BatchNorm tags its moving_mean / moving_var inputs with `__init__` zero / one
    (FSetInputVarAttrOnCompose) and reads them as mutable state (input version 1);
  * unnamed operators are named `<hint><counter>` in creation order (NameManager), the
    Python operators `+ - *` and unary minus map to elemwise_add / elemwise_sub /
    elemwise_mul / _mul_scalar with hints _plus / _minus / _mul / _mulscalar;
  * `tojson` lists the nodes in depth-first post-order from the heads, inputs in order.

tests/test_net_graph.py holds the result to tests/golden/graph.json (extracted from the
reference file by tools/capture_graph.py): same nodes, order, names, attributes, wiring and
heads.  `run()` interprets the inference part of such a graph with NumPy; the same test holds
oracle/net_ref.forward_resnet to it bit for bit, so the float64 oracle the HIP kernels are
checked against computes exactly the graph the reference saved.
"""
import collections

import numpy as np

BN_EPS = 1e-3   # MXNet BatchNorm default eps (no eps attribute in the saved graph)

# parameter inputs an operator creates when they are not passed: (argument names, aux flags)
_OP_INPUTS = {
    "Convolution": ("data", "weight", "bias"),
    "BatchNorm": ("data", "gamma", "beta", "moving_mean", "moving_var"),
    "FullyConnected": ("data", "weight", "bias"),
}
_NAME_HINT = {"elemwise_add": "_plus", "elemwise_sub": "_minus", "elemwise_mul": "_mul", "_mul_scalar": "_mulscalar"}


class Node:
    def __init__(self, op, name, attrs, inputs):
        self.op, self.name, self.attrs, self.inputs = op, name, attrs, inputs   # inputs: [(Node, version)]


class Builder:
    """Symbol composition with MXNet's naming and attribute rules."""

    def __init__(self):
        self.counters = collections.Counter()

    def variable(self, name, shape=None):
        attrs = collections.OrderedDict()
        if shape is not None:
            attrs["__shape__"] = str(tuple(shape))
        return Node("null", name, attrs, [])

    def _auto_name(self, op):
        hint = _NAME_HINT.get(op, op.lower())
        n = self.counters[hint]
        self.counters[hint] += 1
        return "%s%d" % (hint, n)

    def op(self, op, *data, name=None, **kwargs):
        """data: positional symbol inputs; keyword arguments that are Nodes are parameter inputs,
        everything else is an attribute (stringified the way the Python frontend does)."""
        syms = {k: v for k, v in kwargs.items() if isinstance(v, Node)}
        attrs = collections.OrderedDict((k, str(v)) for k, v in sorted(kwargs.items()) if not isinstance(v, Node))
        name = name or self._auto_name(op)
        inputs = []
        arg_names = _OP_INPUTS.get(op)
        if arg_names is None:
            inputs = [(d, 0) for d in data]
        else:
            given = dict(zip(arg_names, data))
            given.update(syms)
            for i, arg in enumerate(arg_names):
                node = given.get(arg)
                if node is None:            # created on compose: inherits the operator's attributes
                    node = Node("null", "%s_%s" % (name, arg), collections.OrderedDict(attrs), [])
                version = 0
                if op == "BatchNorm" and i >= 3:
                    node.attrs.setdefault("__init__", '["zero", {}]' if i == 3 else '["one", {}]')
                    version = 1             # mutable auxiliary state
                inputs.append((node, version))
        return Node(op, name, attrs, inputs)


def _conv_act(b, data, num_filter, kernel, name):
    """policy_value_net_mxnet.py:41-60 (dobn=True, act='relu')."""
    pad = (int(kernel[0] / 2), int(kernel[1] / 2))
    w, bias = b.variable(name + "_weight"), b.variable(name + "_bias")
    conv = b.op("Convolution", data, weight=w, bias=bias, num_filter=num_filter, kernel=kernel, pad=pad, name=name)
    bn = b.op("BatchNorm", conv, gamma=b.variable(name + "_gamma"), beta=b.variable(name + "_beta"),
              moving_mean=b.variable(name + "_mean"), moving_var=b.variable(name + "_var"), name=name + "_bn")
    return b.op("Activation", bn, act_type="relu", name=name + "_act")


def _fc_self(b, data, num_hidden, name):
    """policy_value_net_mxnet.py:62-68."""
    return b.op("FullyConnected", data, weight=b.variable(name + "_weight"), bias=b.variable(name + "_bias"),
                num_hidden=num_hidden, name=name)


def backbone_resnet(b, input_states, n_blocks, n_filter, hw):
    """policy_value_net_mxnet.py:70-102 -> (action_1, evaluation)."""
    net = _conv_act(b, input_states, 128, (3, 3), "res_conv1")
    for i in range(1, n_blocks + 1):
        pre = net
        net = b.op("Convolution", net, name="convA%d" % i, kernel=(3, 3), pad=(1, 1), num_filter=n_filter)
        net = b.op("BatchNorm", net, name="bnA%d" % i, fix_gamma=False)
        net = b.op("Activation", net, name="actA%d" % i, act_type="relu")
        net = b.op("Convolution", net, name="convB%d" % i, kernel=(3, 3), pad=(1, 1), num_filter=n_filter)
        net = b.op("BatchNorm", net, name="bnB%d" % i, fix_gamma=False)
        net = b.op("elemwise_add", net, pre)
        net = b.op("Activation", net, name="actB%d" % i, act_type="relu")
    pol = _conv_act(b, net, 4, (1, 1), "conv3_1_1")
    pol = b.op("Dropout", b.op("Flatten", pol), p=0.5)
    action = b.op("SoftmaxActivation", _fc_self(b, pol, hw, "fc_3_1_1"), name="Act_SILER")
    val = _conv_act(b, net, 2, (1, 1), "conv3_2_1")
    val = b.op("Dropout", b.op("Flatten", val), p=0.5)
    evaluation = b.op("Activation", _fc_self(b, val, 1, "fc_3_2_1"), act_type="tanh")
    return action, evaluation


def train_graph(batch=128, c_in=9, height=15, width=15, n_blocks=10, n_filter=128):
    """policy_value_net_mxnet.py:173-194: the group [MakeLoss(value + policy loss), MakeLoss(BlockGrad(entropy))]."""
    b = Builder()
    hw = height * width
    states = b.variable("input_states", (batch, c_in, height, width))
    action, evaluation = backbone_resnet(b, states, n_blocks, n_filter, hw)
    probs = b.variable("mcts_probs", (batch, hw))
    policy_loss = b.op("_mul_scalar", b.op("sum", b.op("elemwise_mul", b.op("log", action), probs), axis=1), scalar=-1.0)
    policy_loss = b.op("mean", policy_loss)
    labels = b.variable("input_labels", (batch, 1))
    value_loss = b.op("mean", b.op("square", b.op("elemwise_sub", labels, evaluation)))
    loss = b.op("MakeLoss", b.op("elemwise_add", value_loss, policy_loss))
    entropy = b.op("sum", b.op("elemwise_mul", b.op("_mul_scalar", action, scalar=-1.0), b.op("log", action)), axis=1)
    entropy = b.op("MakeLoss", b.op("BlockGrad", b.op("mean", entropy)))
    return [loss, entropy]


def serialise(heads):
    """MXNet's tojson order: depth-first post-order from the heads, inputs in order.
    -> {"nodes": [{op, name, attrs, inputs: [[node index, 0, version]]}], "arg_nodes": [...], "heads": [...]}."""
    index, nodes = {}, []

    def visit(node):
        if id(node) in index:
            return
        for inp, _ in node.inputs:
            visit(inp)
        index[id(node)] = len(nodes)
        nodes.append(node)

    for h in heads:
        visit(h)
    out = []
    for nd in nodes:
        out.append({"op": nd.op, "name": nd.name, "attrs": dict(nd.attrs),
                    "inputs": [[index[id(i)], 0, v] for i, v in nd.inputs]})
    return {"nodes": out, "arg_nodes": [i for i, nd in enumerate(nodes) if nd.op == "null"],
            "heads": [[index[id(h)], 0, 0] for h in heads]}


def infer_arg_shapes(graph):
    """Shapes of the variables of a serialised graph, from the `__shape__` of its inputs and the
    operator attributes (what Module.bind infers): {name: shape}."""
    nodes = graph["nodes"]
    shape = {}

    def tup(s):
        return tuple(int(x) for x in s.strip("()").split(",") if x.strip())

    out = {}
    for i, nd in enumerate(nodes):
        a, ins = nd["attrs"], [j for j, _, _ in nd["inputs"]]
        if nd["op"] == "null":
            if "__shape__" in a:
                shape[i] = tup(a["__shape__"])
            continue
        x = shape.get(ins[0]) if ins else None
        if nd["op"] == "Convolution":
            k, nf = tup(a["kernel"]), int(a["num_filter"])
            pad = tup(a.get("pad", "(0, 0)"))
            shape[ins[1]], shape[ins[2]] = (nf, x[1]) + k, (nf,)
            shape[i] = (x[0], nf, x[2] + 2 * pad[0] - k[0] + 1, x[3] + 2 * pad[1] - k[1] + 1)
        elif nd["op"] == "BatchNorm":
            for j in ins[1:]:
                shape[j] = (x[1],)
            shape[i] = x
        elif nd["op"] == "FullyConnected":
            nh, flat = int(a["num_hidden"]), int(np.prod(x[1:]))
            shape[ins[1]], shape[ins[2]] = (nh, flat), (nh,)
            shape[i] = (x[0], nh)
        elif nd["op"] == "Flatten":
            shape[i] = (x[0], int(np.prod(x[1:])))
        elif nd["op"] in ("sum",):
            shape[i] = (x[0],)
        elif nd["op"] in ("mean",):
            shape[i] = (1,)
        else:                               # elementwise / activation / dropout / loss wrappers
            shape[i] = x
    for i, nd in enumerate(nodes):
        if nd["op"] == "null":
            out[nd["name"]] = shape[i]
    return out


def _conv(x, w, b, pad):
    n, ci, h, wd = x.shape
    co, _, kh, kw = w.shape
    xp = np.zeros((n, ci, h + 2 * pad[0], wd + 2 * pad[1]), dtype=x.dtype)
    xp[:, :, pad[0]:pad[0] + h, pad[1]:pad[1] + wd] = x
    out = np.zeros((n, co, h + 2 * pad[0] - kh + 1, wd + 2 * pad[1] - kw + 1), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, :, ky:ky + out.shape[2], kx:kx + out.shape[3]]
            out += np.einsum("nchw,oc->nohw", patch, w[:, :, ky, kx], optimize=True)
    return out + b[None, :, None, None]


def run(graph, prm, planes, outputs=("Act_SILER", "activation0"), dtype=np.float64):
    """Interpret the inference part of a serialised graph (MXNet-1.x operator semantics as in
    oracle/net_ref.py: BatchNorm with moving statistics and eps 1e-3, gamma := 1 unless
    fix_gamma is 'False'; Dropout = identity).  -> {node name: array} for `outputs` plus the
    pre-activation inputs of both heads ('fc_3_1_1', 'fc_3_2_1')."""
    nodes = graph["nodes"]
    want = set(outputs) | {"fc_3_1_1", "fc_3_2_1"}
    need, stack = set(), [i for i, nd in enumerate(nodes) if nd["name"] in want]
    while stack:
        i = stack.pop()
        if i in need:
            continue
        need.add(i)
        stack.extend(j for j, _, _ in nodes[i]["inputs"])
    val = {}
    for i, nd in enumerate(nodes):
        if i not in need:
            continue
        op, a = nd["op"], nd["attrs"]
        ins = [val[j] for j, _, _ in nd["inputs"]]
        if op == "null":
            val[i] = np.asarray(planes if nd["name"] == "input_states" else prm[nd["name"]]).astype(dtype)
        elif op == "Convolution":
            pad = tuple(int(t) for t in a.get("pad", "(0, 0)").strip("()").split(","))
            val[i] = _conv(ins[0], ins[1], ins[2], pad)
        elif op == "BatchNorm":
            x, gamma, beta, mean, var = ins
            if a.get("fix_gamma", "True") != "False":
                gamma = np.ones_like(gamma)
            sh = (1, -1, 1, 1)
            inv = (1.0 / np.sqrt(var + dtype(BN_EPS))).astype(dtype)
            val[i] = (x - mean.reshape(sh)) * inv.reshape(sh) * gamma.reshape(sh) + beta.reshape(sh)
        elif op == "Activation":
            val[i] = np.maximum(ins[0], 0) if a["act_type"] == "relu" else np.tanh(ins[0])
        elif op == "elemwise_add":
            val[i] = ins[0] + ins[1]
        elif op == "Flatten":
            val[i] = ins[0].reshape(ins[0].shape[0], -1)
        elif op == "Dropout":
            val[i] = ins[0]
        elif op == "FullyConnected":
            val[i] = ins[0] @ ins[1].T + ins[2]
        elif op == "SoftmaxActivation":
            e = np.exp(ins[0] - ins[0].max(axis=1, keepdims=True))
            val[i] = e / e.sum(axis=1, keepdims=True)
        else:
            raise ValueError("operator %s is not part of the inference graph" % op)
    return {nodes[i]["name"]: v for i, v in val.items() if nodes[i]["name"] in want}
