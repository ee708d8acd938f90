"""Training step on the HIP kernels: the consumer of the self-play path's tuples.

Everything the reference defines around the optimiser step (policy_value_net_mxnet.py:173-212, :282-299 and
train_mxnet.py:194-240) restated as an explicit forward pass, an explicit backward pass over the saved activations
and one Adam launch -- each operator a hand-written HIP kernel behind include/alphapig_hip.h (alphapig_amd/hipconv.py
holds the one-line wrappers).  No autograd and no PyTorch arithmetic: torch tensors are device buffers here, as in
the self-play path.  The module needs the HIP library and a GPU; without them construction raises.

  graph     training-mode BatchNorm (batch statistics, eps 1e-3, momentum 0.9, gamma frozen at 1 where the
            reference leaves fix_gamma at MXNet's default), Dropout(0.5) on both flattened head inputs (the training
            graph shares create_backbone_resnet), softmax / tanh heads
  loss      mean((z - v)^2) + mean(-sum(pi * log p, axis=1)); entropy monitor mean(sum(-p log p))
  update    MXNet Adam as Module.init_optimizer configures it: g = grad / batch_size + wd * w (rescale_grad =
            1/batch_size on top of the mean loss; wd = 1e-4 on *_weight and *_gamma only), m/v moments,
            lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t), eps 1e-8
  policy_update   epochs x train_step with the KL-adaptive learning-rate multiplier and the 4 * kl_targ early stop

The 10-block / 128-filter / 15x15 network keeps its trunk activations in the self-play kernels' padded-row layout
[n][128][15][16] from the stem's BatchNorm to the heads (forward and data gradient on the fused Winograd kernel of
the self-play path, weight gradient through the Winograd domain); other shapes (the 8x8 simple net, other filter
counts) run dense on the direct MFMA kernel.

PARITY UNPINNED like the forward's numbers (MXNet absent, no recorded training runs).  Checked on the GPU against
tests/torch_trainer.py (the same graph in PyTorch float64 / float32 autograd), which tests/test_train.py in turn
checks against oracle/train_ref.py (NumPy float64 loss + finite differences, NumPy Adam).
"""
import collections

import numpy as np

BN_EPS = 1e-3
BN_MOMENTUM = 0.9
SIMPLE_CONVS = ("conv1", "conv2", "conv3", "conv4", "conv5", "conv_final")
DeviceBatch = collections.namedtuple("DeviceBatch", "states pis zs")     # HipTrainer.upload()


class HipTrainer(object):
    def __init__(self, params, net_kind="resnet", n_blocks=10, batch_size=512, wd=1e-4, device_index=0, dropout=0.5,
                 seed=0, height=None, width=None, dropout_step0=0):
        """params: name -> array, in any order.  height / width: the board (default: square, from the policy head's size).
        seed / dropout_step0: the dropout masks are a stateless hash of (seed, dropout_step0 + step, element) -- a trainer
        that is re-created (set_params, a resumed checkpoint) or one of several ranks passes its own seed / the steps
        already taken, or it replays the mask sequence of a fresh trainer."""
        import torch
        from . import _native, hipconv
        if not torch.cuda.is_available():
            raise RuntimeError("HipTrainer needs a GPU: the training step has no CPU path")
        _native.hip()                          # raises when libalphapig_hip.so is missing
        self.torch, self.ops = torch, hipconv
        self.kind, self.n_blocks = net_kind, n_blocks
        self.batch_size, self.wd, self.dropout, self.seed = batch_size, wd, dropout, seed
        self.device = torch.device("cuda", device_index)
        self.p = collections.OrderedDict()
        for k, v in params.items():
            self.p[k] = torch.tensor(np.ascontiguousarray(v, dtype=np.float32), device=self.device)
        stem = "res_conv1_weight" if net_kind == "resnet" else "conv1_weight"
        if stem not in self.p:
            raise ValueError("parameter %s missing (net_kind %r)" % (stem, net_kind))
        self.c_in = int(self.p[stem].shape[1])
        self.hw = int(self.p["fc_3_1_1_bias"].shape[0])
        self.side = int(round(self.hw ** 0.5))
        if (height is None) != (width is None):
            raise ValueError("pass both height and width, or neither")
        if height is not None and (int(height) * int(width) != self.hw or int(height) != int(width)):
            raise ValueError("the HIP training kernels take square boards whose size matches the policy head (%d cells)" % self.hw)
        if self.side * self.side != self.hw:
            raise ValueError("policy head of %d cells is not a square board" % self.hw)
        self.dropout_step0 = int(dropout_step0)
        self.stat_names = [k for k in self.p if k.endswith(("_mean", "_var", "_moving_mean", "_moving_var"))]
        # gammas of the fix_gamma BatchNorm layers (every conv_act layer: res_conv1, the two 1x1 heads, and all of the
        # simple net; policy_value_loss.json nodes 9 / 208 / 226 carry no fix_gamma=False) never enter the graph.
        # MXNet rewrites them to 1 on every forward, so they are pinned to 1 here and kept out of the optimiser --
        # left in, weight decay + Adam move them by ~lr per step for ever and the drift lands in saved checkpoints.
        self.fixed_gamma_names = [k for k in self.p if k.endswith("_gamma") and not k.startswith(("bnA", "bnB"))]
        for k in self.fixed_gamma_names:
            self.p[k].fill_(1.0)
        self.train_names = [k for k in self.p if k not in self.stat_names and k not in self.fixed_gamma_names]
        self.m = {k: torch.zeros_like(self.p[k]) for k in self.train_names}
        self.v = {k: torch.zeros_like(self.p[k]) for k in self.train_names}
        self.t = 0
        self.grad = {}
        # the self-play kernels' padded-row layout for the trunk when the net has their shape
        self.rows16 = (net_kind == "resnet" and n_blocks > 0 and self.side == 15 and
                       tuple(self.p["convA1_weight"].shape) == (128, 128, 3, 3))
        # ... whose 2 n_blocks weight tensors then live back to back in one buffer (self.p holds views): one launch packs
        # all of them for the Winograd kernel in both orientations at the start of a step (hipconv.wino_pack_many)
        self._trunk_names = []
        self._trunk_w = self._upk = None
        if self.rows16:
            self._trunk_names = ["conv%s%d_weight" % (ab, i) for i in range(1, n_blocks + 1) for ab in "AB"]
            self._trunk_w = torch.empty((len(self._trunk_names), 128, 128, 3, 3), dtype=torch.float32, device=self.device)
            for j, k in enumerate(self._trunk_names):
                self._trunk_w[j].copy_(self.p[k])
                self.p[k] = self._trunk_w[j]
        self._eval = None
        self._eval_t = -1
        self.tape = None

    # ---- forward: every activation the backward pass reads goes on the tape -----------------------------------------
    def _conv_act_fwd(self, x, name, layout):
        """conv_act of the reference (policy_value_net_mxnet.py:28-39): Convolution -> BatchNorm(fix_gamma) -> relu"""
        o, p = self.ops, self.p
        w = p[name + "_weight"]
        if w.shape[2] == 1:
            y = o.conv1x1_fwd(x, w, p[name + "_bias"], layout)
            blay = o.DENSE
        else:
            y = o.conv3x3_fwd(x, w, p[name + "_bias"], layout)
            blay = layout
        a, mean, invstd = o.bn_fwd(y, None, p[name + "_beta"], p[name + "_mean"], p[name + "_var"], None, True, blay,
                                   1.0 - BN_MOMENTUM, BN_EPS)
        return a, (name, x, y, a, mean, invstd, layout)

    def _conv_act_bwd(self, da, rec, need_dx, dx_acc=None):
        o, p, g = self.ops, self.p, self.grad
        name, x, y, a, mean, invstd, layout = rec
        w = p[name + "_weight"]
        one = w.shape[2] == 1
        dy, _, _, g[name + "_beta"] = o.bn_bwd(da, y, a, None, mean, invstd, True, False, o.DENSE if one else layout)
        if one:
            dx, g[name + "_weight"], g[name + "_bias"] = o.conv1x1_bwd(x, w, dy, layout, dx_acc)
            return dx
        g[name + "_weight"] = o.conv3x3_wgrad(x, dy, layout)
        g[name + "_bias"] = o.bias_grad(dy, layout)
        return o.conv3x3_dgrad(dy, w, layout) if need_dx else None

    def _forward(self, states, step):
        o, p = self.ops, self.p
        tape = {"blocks": [], "stack": []}
        if self.kind == "resnet":
            x, rec = self._conv_act_fwd(states, "res_conv1", o.DENSE)
            tape["stem"] = rec
            lay = o.ROWS16 if self.rows16 else o.DENSE
            upk = None
            if self.rows16:
                x = o.to_rows16(x)
                self._upk = upk = o.wino_pack_many(self._trunk_w, self._upk)
            tape["upk"] = upk
            for i in range(1, self.n_blocks + 1):
                A, B = "A%d" % i, "B%d" % i
                ua, ub = (upk[2 * i - 2, 0], upk[2 * i - 1, 0]) if upk is not None else (None, None)
                sa = sb = None
                fused = upk is not None and int(x.shape[0]) <= 16384       # (apz_wino_conv_stats: one launch)
                if fused:               # the Winograd kernel's epilogue leaves the BatchNorm's per-board sums: no statistics pass
                    ya, sa = o.conv3x3_fwd_stats(x, p["conv" + A + "_weight"], p["conv" + A + "_bias"], upk=ua)
                else:
                    ya = o.conv3x3_fwd(x, p["conv" + A + "_weight"], p["conv" + A + "_bias"], lay, upk=ua)
                wm = self.rows16        # padded rows: the ReLU decisions as a byte per four elements for the backward pass
                ha, ma, ia, *ka = o.bn_fwd(ya, p["bn" + A + "_gamma"], p["bn" + A + "_beta"], p["bn" + A + "_moving_mean"],
                                           p["bn" + A + "_moving_var"], None, True, lay, 1.0 - BN_MOMENTUM, BN_EPS, stats=sa,
                                           want_mask=wm)
                if fused:
                    yb, sb = o.conv3x3_fwd_stats(ha, p["conv" + B + "_weight"], p["conv" + B + "_bias"], upk=ub)
                else:
                    yb = o.conv3x3_fwd(ha, p["conv" + B + "_weight"], p["conv" + B + "_bias"], lay, upk=ub)
                out, mb, ib, *kb = o.bn_fwd(yb, p["bn" + B + "_gamma"], p["bn" + B + "_beta"], p["bn" + B + "_moving_mean"],
                                            p["bn" + B + "_moving_var"], x, True, lay, 1.0 - BN_MOMENTUM, BN_EPS, stats=sb,
                                            want_mask=wm)
                tape["blocks"].append((x, ya, ha, ma, ia, yb, out, mb, ib, ka[0] if ka else None, kb[0] if kb else None))
                x = out
        else:
            lay = o.DENSE
            x = states
            for name in SIMPLE_CONVS:
                x, rec = self._conv_act_fwd(x, name, lay)
                tape["stack"].append(rec)
        tape["layout"] = lay
        n = int(x.shape[0])
        pol, tape["pol"] = self._conv_act_fwd(x, "conv3_1_1", lay)
        val, tape["val"] = self._conv_act_fwd(x, "conv3_2_1", lay)
        pol, val = pol.view(n, -1), val.view(n, -1)
        if self.dropout > 0:
            keep = 1.0 - self.dropout
            pol = o.dropout(pol, keep, self.seed, 2 * (self.dropout_step0 + step))
            val = o.dropout(val, keep, self.seed, 2 * (self.dropout_step0 + step) + 1)
        tape["pol_in"], tape["val_in"] = pol, val
        logits = o.fc_fwd(pol, p["fc_3_1_1_weight"], p["fc_3_1_1_bias"])
        vlogit = o.fc_fwd(val, p["fc_3_2_1_weight"], p["fc_3_2_1_bias"]).view(n)
        return logits, vlogit, tape

    # ---- backward ---------------------------------------------------------------------------------------------------
    def _backward(self, tape, dlogits, dvlogit):
        o, p, g = self.ops, self.p, self.grad
        n = int(dlogits.shape[0])
        lay = tape["layout"]
        dpol, g["fc_3_1_1_weight"], g["fc_3_1_1_bias"] = o.fc_bwd(tape["pol_in"], p["fc_3_1_1_weight"], dlogits)
        dval, g["fc_3_2_1_weight"], g["fc_3_2_1_bias"] = o.fc_bwd(tape["val_in"], p["fc_3_2_1_weight"], dvlogit.view(n, 1))
        if self.dropout > 0:
            keep = 1.0 - self.dropout
            dpol = o.dropout(dpol, keep, self.seed, 2 * (self.dropout_step0 + tape["step"]))
            dval = o.dropout(dval, keep, self.seed, 2 * (self.dropout_step0 + tape["step"]) + 1)
        # the two heads share their input: both BatchNorm backward passes, then ONE pass over the trunk output for dx
        dys = []
        for da, rec in ((dpol.view(n, 4, self.side, self.side), tape["pol"]), (dval.view(n, 2, self.side, self.side), tape["val"])):
            name, _, y, a, mean, invstd, _ = rec
            dy, _, _, g[name + "_beta"] = o.bn_bwd(da, y, a, None, mean, invstd, True, False, o.DENSE)
            dys.append(dy)
        xh = tape["pol"][1]
        dx, g["conv3_1_1_weight"], g["conv3_1_1_bias"], g["conv3_2_1_weight"], g["conv3_2_1_bias"] = o.conv1x1_bwd_pair(
            xh, p["conv3_1_1_weight"], dys[0], p["conv3_2_1_weight"], dys[1], tape["pol"][6])
        if self.kind == "resnet":
            upk = tape.get("upk")
            # the trunk convolutions' bias gradients = column sums of the dx their BatchNorms hand back: every bn_bwd
            # leaves its per-split sums in its own columns of ONE matrix, added after the loop in one launch
            nb = self.n_blocks
            nf = int(tape["blocks"][0][1].shape[1]) if nb else 0
            parts = o._empty((o.bn_bwd_splits(tape["blocks"][0][1], lay), 2 * nb * nf), dx) if nb else None
            for i in range(nb, 0, -1):
                A, B = "A%d" % i, "B%d" % i
                x, ya, ha, ma, ia, yb, out, mb, ib, ka, kb = tape["blocks"][i - 1]
                ua, ub = (upk[2 * i - 2, 1], upk[2 * i - 1, 1]) if upk is not None else (None, None)
                ca, cb = (2 * i - 2) * nf, (2 * i - 1) * nf
                dyb, dskip, g["bn" + B + "_gamma"], g["bn" + B + "_beta"] = o.bn_bwd(dx, yb, out, p["bn" + B + "_gamma"], mb, ib,
                                                                                    True, True, lay, dxsum=parts[:, cb:cb + nf], mask=kb)
                g["conv" + B + "_weight"] = o.conv3x3_wgrad(ha, dyb, lay)
                dha = o.conv3x3_dgrad(dyb, p["conv" + B + "_weight"], lay, upk=ub)
                dya, _, g["bn" + A + "_gamma"], g["bn" + A + "_beta"] = o.bn_bwd(dha, ya, ha, p["bn" + A + "_gamma"], ma, ia,
                                                                                True, False, lay, dxsum=parts[:, ca:ca + nf], mask=ka)
                g["conv" + A + "_weight"] = o.conv3x3_wgrad(x, dya, lay)
                dx = o.conv3x3_dgrad(dya, p["conv" + A + "_weight"], lay, add=dskip, upk=ua)   # trunk + skip gradients meet
            db = o.colsum(parts) if nb else None
            for i in range(1, nb + 1):
                g["convA%d_bias" % i] = db[(2 * i - 2) * nf:(2 * i - 1) * nf]
                g["convB%d_bias" % i] = db[(2 * i - 1) * nf:2 * i * nf]
            if self.rows16:
                dx = o.from_rows16(dx)
            self._conv_act_bwd(dx, tape["stem"], False)
        else:
            for j in range(len(SIMPLE_CONVS) - 1, -1, -1):
                dx = self._conv_act_bwd(dx, tape["stack"][j], j > 0)

    # ---- one optimiser step (policy_value_net_mxnet.py:282-299) -----------------------------------------------------
    def _to(self, a, shape):
        t = self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).reshape(shape)
        return t.to(self.device).contiguous()

    def upload(self, state_batch, mcts_probs, winner_batch):
        """-> a DeviceBatch that train_step / loss_and_grads take in place of the three host arrays: a policy_update feeds
        the SAME mini-batch to every epoch (train_mxnet.py:201-207), so it goes to the device once (4.6 MB at batch 512:
        0.26 ms of an idle GPU in front of every step otherwise)."""
        return DeviceBatch(self._to(state_batch, (-1, self.c_in, self.side, self.side)), self._to(mcts_probs, (-1, self.hw)),
                           self._to(winner_batch, (-1,)))

    def loss_and_grads(self, state_batch, mcts_probs=None, winner_batch=None, keep_tape=False):
        """Forward + backward in training mode (moving statistics are updated).  -> (loss3 device tensor =
        (value loss, policy loss, entropy)); the gradients of the MEAN loss land in self.grad.  keep_tape: hold on to
        the saved activations afterwards (relu_masks(); tests).  state_batch: host array, or upload()'s DeviceBatch."""
        if isinstance(state_batch, DeviceBatch):
            states, pis, zs = state_batch
        else:
            states, pis, zs = self.upload(state_batch, mcts_probs, winner_batch)
        self.grad = {}
        logits, vlogit, tape = self._forward(states, self.t)
        tape["step"] = self.t
        out = self.ops.pv_loss(logits, vlogit, pis, zs, grads=True)
        self._backward(tape, out["dlogits"], out["dvlogit"])
        self.tape = tape if keep_tape else None
        return out["loss3"]

    def relu_masks(self):
        """{layer: [n][C][H][W] bool} -- which activations of the kept forward pass were positive.  (A comparator that
        takes its ReLU decisions from here differs from this trainer by rounding only: an activation within rounding
        distance of zero otherwise lands on different sides in different arithmetic and moves whole gradient rows.)"""
        cut = (lambda t: t[..., :15]) if self.rows16 else (lambda t: t)
        tape, out = self.tape, {}
        recs = [tape["pol"], tape["val"]] + tape["stack"] + ([tape["stem"]] if "stem" in tape else [])
        for rec in recs:
            out[rec[0]] = rec[3] > 0
        for i, blk in enumerate(tape["blocks"], 1):
            out["bnA%d" % i] = cut(blk[2]) > 0
            out["block%d" % i] = cut(blk[6]) > 0
        return out

    def train_step(self, state_batch, mcts_probs, winner_batch, learning_rate, keep_tape=False):
        """One optimiser step -> (loss, entropy).  state_batch may be upload()'s DeviceBatch (mcts_probs / winner_batch are
        then ignored)."""
        loss3 = self.loss_and_grads(state_batch, mcts_probs, winner_batch, keep_tape)
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        lr_t = learning_rate * (1.0 - b2 ** self.t) ** 0.5 / (1.0 - b1 ** self.t)
        entries = [(self.p[k], self.grad[k], self.m[k], self.v[k], self.wd if k.endswith(("_weight", "_gamma")) else 0.0)
                   for k in self.train_names]
        self.ops.adam_step(entries, lr_t, b1, b2, eps, 1.0 / self.batch_size, self.device)
        l3 = loss3.cpu().numpy()           # the step's only device -> host copy (12 bytes), and its synchronisation
        return float(l3[0] + l3[1]), float(l3[2])

    def sync_evaluator(self, net):
        """Give `net` (a PolicyValueNet of the same architecture) this trainer's current weights, device to device:
        folding and packing run as kernels behind the optimiser step on the same stream (apz_load_weights_dev)."""
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        net.load_device_params(self.p, stream)

    def policy_value(self, state_batch):
        """Inference-mode (moving statistics) probabilities and values for the KL monitor: the self-play path's own
        evaluator (PolicyValueNet) on the current weights."""
        from .policy_value_net import PolicyValueNet
        if self._eval is None:
            nf = int(next(iter(self.p.values())).shape[0]) if self.kind == "resnet" else 128
            self._eval = PolicyValueNet(self.side, self.side, batch_size=self.batch_size, n_blocks=self.n_blocks, n_filter=nf,
                                        model_params=self.get_params(), net_kind=self.kind, c_in=self.c_in,
                                        device=self.device.index or 0)
            self._eval_t = self.t
        elif self._eval_t != self.t:
            self.sync_evaluator(self._eval)
            self._eval_t = self.t
        return self._eval.policy_value(state_batch)

    def get_params(self):
        return collections.OrderedDict((k, v.cpu().numpy()) for k, v in self.p.items())

    def get_grads(self):
        return collections.OrderedDict((k, self.grad[k].cpu().numpy()) for k in self.train_names)

    def close(self):
        if self._eval is not None:
            self._eval.close()
            self._eval = None


def explained_variance(winner_z, value):
    """train_mxnet.py:222-227: 1 - Var(z - v) / Var(z) (NumPy semantics: a constant z gives nan / -inf, as there)."""
    z = np.asarray(winner_z, dtype=np.float64)
    v = np.asarray(value, dtype=np.float64).reshape(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(1.0 - np.var(z - v) / np.var(z))


def policy_update(trainer, mini_batch, learn_rate=1e-3, lr_multiplier=1.0, epochs=8, kl_targ=0.02, evaluator=None, monitors=None):
    """train_mxnet.py:194-240: epochs x train_step with KL early stop and the adaptive LR multiplier.
    mini_batch: list of (state, mcts_prob, winner_z).  `evaluator.policy_value(states)` (the HIP
    PolicyValueNet) supplies old/new predictions when given, else the trainer's own inference
    graph.  -> (loss, entropy, kl, lr_multiplier); `monitors` (a dict, optional) receives the reference's value-head
    monitors explained_var_old / explained_var_new (train_mxnet.py:222-227) and the learning rate used."""
    states = np.stack([np.ascontiguousarray(d[0]) for d in mini_batch]).astype(np.float32)
    pis = np.stack([d[1] for d in mini_batch]).astype(np.float32)
    zs = np.array([d[2] for d in mini_batch], dtype=np.float32)
    pv = (evaluator.policy_value if evaluator is not None else trainer.policy_value)
    old_probs, old_v = pv(states)
    new_v = old_v
    loss = entropy = kl = 0.0
    batch = trainer.upload(states, pis, zs) if hasattr(trainer, "upload") else states     # once for all epochs
    lr_used = lr_multiplier                               # (the reference logs learn_rate * the multiplier the epochs ran with)
    for _ in range(epochs):
        loss, entropy = trainer.train_step(batch, pis, zs, learn_rate * lr_multiplier)
        if evaluator is not None:
            if hasattr(trainer, "sync_evaluator") and hasattr(evaluator, "load_device_params"):
                trainer.sync_evaluator(evaluator)
            else:
                evaluator.set_params(trainer.get_params())
        new_probs, new_v = pv(states)
        kl = float(np.mean(np.sum(old_probs * (np.log(old_probs + 1e-10) - np.log(new_probs + 1e-10)), axis=1)))
        if kl > kl_targ * 4:
            break
    if kl > kl_targ * 2 and lr_multiplier > 0.05:        # train_mxnet.py:215-218
        lr_multiplier /= 1.5
    elif kl < kl_targ / 2 and lr_multiplier < 20:
        lr_multiplier *= 1.5
    if monitors is not None:
        monitors["explained_var_old"] = explained_variance(zs, old_v)
        monitors["explained_var_new"] = explained_variance(zs, new_v)
        monitors["learn_rate"] = float(learn_rate * lr_used)
    return loss, entropy, kl, lr_multiplier
