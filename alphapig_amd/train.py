"""Training step -- INTERIM implementation of the first "next" row.

The self-play hot path (this repository's scope) is hand-written HIP; the training step that
consumes its tuples is, for now, PyTorch-ROCm autograd over the same MXNet-named parameter dict
(SURVEY.md 8f explicitly allows that interim).  What IS restated by hand is everything the
reference defines around the backward pass (reference policy_value_net_mxnet.py:173-212,
:282-299 and train_mxnet.py:194-240):

  loss      mean((z - v)^2) + mean(-sum(pi * log p, axis=1)); entropy monitor mean(sum(-p log p))
  graph     training-mode BatchNorm (batch statistics, eps 1e-3, momentum 0.9, gamma frozen at 1
            where the reference leaves fix_gamma at MXNet's default), Dropout(0.5) on both
            flattened head inputs (the training graph shares create_backbone_resnet)
  update    MXNet Adam as Module.init_optimizer configures it: g = grad / batch_size + wd * w
            (rescale_grad = 1/batch_size on top of the mean loss; wd = 1e-4 on *_weight and
            *_gamma only), m/v moments, lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t), eps 1e-8
  policy_update   epochs x train_step with the KL-adaptive learning-rate multiplier and the
            4 * kl_targ early stop

PARITY UNPINNED like the forward (MXNet absent, no recorded training runs); checked against
oracle/train_ref.py (NumPy float64 loss + finite differences, NumPy Adam) in tests/test_train.py.
"""
import collections

import numpy as np

BN_EPS = 1e-3
BN_MOMENTUM = 0.9


def _torch():
    import torch
    return torch


class TorchTrainer(object):
    def __init__(self, params, net_kind="resnet", n_blocks=10, batch_size=512, wd=1e-4, device=None,
                 dtype=None, dropout=0.5, seed=0, conv_backend=None, trunk_backend=None):
        torch = _torch()
        self.torch = torch
        self.kind, self.n_blocks = net_kind, n_blocks
        self.batch_size, self.wd, self.dropout = batch_size, wd, dropout
        # "hip": 3x3 convolutions (forward, dgrad, wgrad), the trunk's BatchNorm and Adam on this repository's
        # kernels (hipconv.py) -- the default on a GPU; "torch": everything in PyTorch (the CPU path, and the
        # reference graph the HIP path is tested against)
        self.conv_backend = conv_backend
        # "hip16": the residual trunk end to end on HIP kernels in the self-play path's padded-row layout --
        # Winograd pair kernel for forward / data gradient (no layout copies), weight-gradient kernel on the same
        # layout, BatchNorm (+ residual) + ReLU forward and backward (hipconv.bn_act).  Default: on whenever the
        # 3x3 convolutions are on HIP and the net is the 15x15 / 128-filter one; trunk_backend="torch" turns it off.
        self.trunk_backend = trunk_backend
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        self.dtype = dtype or torch.float32
        if self.conv_backend is None:
            self.conv_backend = "hip" if (self.device.type == "cuda" and self.dtype == torch.float32) else "torch"
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.p = collections.OrderedDict()
        for k, v in params.items():
            t = torch.tensor(np.asarray(v), dtype=self.dtype, device=self.device)
            self.p[k] = t
        self.stat_names = [k for k in self.p if k.endswith(("_mean", "_var", "_moving_mean", "_moving_var"))]
        # gammas of the fix_gamma BatchNorm layers (every conv_act layer: res_conv1, the two 1x1 heads, and all of the
        # simple net; policy_value_loss.json nodes 9 / 208 / 226 carry no fix_gamma=False) never enter the graph.
        # MXNet rewrites them to 1 on every forward, so they are pinned to 1 here and kept out of the optimiser --
        # left in, weight decay + Adam move them by ~lr per step for ever and the drift lands in saved checkpoints.
        self.fixed_gamma_names = [k for k in self.p if k.endswith("_gamma") and not k.startswith(("bnA", "bnB"))]
        for k in self.fixed_gamma_names:
            self.p[k].fill_(1.0)
        self.train_names = [k for k in self.p if k not in self.stat_names and k not in self.fixed_gamma_names]
        for k in self.train_names:
            self.p[k].requires_grad_(True)
        self.m = {k: torch.zeros_like(self.p[k]) for k in self.train_names}
        self.v = {k: torch.zeros_like(self.p[k]) for k in self.train_names}
        self.t = 0

    # ---- graph ------------------------------------------------------------------------------
    def _bn(self, x, name, fix_gamma, mean_n, var_n, train):
        F = self.torch.nn.functional
        gamma = None if fix_gamma else self.p[name + "_gamma"]
        if gamma is None:
            gamma = self.torch.ones_like(self.p[name + "_beta"])
        rm, rv = self.p[name + mean_n], self.p[name + var_n]
        return F.batch_norm(x, rm, rv, gamma, self.p[name + "_beta"], training=train,
                            momentum=1.0 - BN_MOMENTUM, eps=BN_EPS)

    def _conv(self, x, w, b, k):
        if k == 3 and self.conv_backend == "hip":
            from . import hipconv
            if hipconv.supported(x, w):
                return hipconv.conv3x3(x, w, b)
        return self.torch.nn.functional.conv2d(x, w, b, padding=k // 2)

    def _conv_act(self, x, name, k, train):
        F = self.torch.nn.functional
        y = self._conv(x, self.p[name + "_weight"], self.p[name + "_bias"], k)
        return F.relu(self._bn(y, name, True, "_mean", "_var", train))

    def _use_hip16(self, x, train):
        if self.trunk_backend == "torch" or self.conv_backend != "hip" or not train or self.kind != "resnet":
            return False
        return (x.is_cuda and tuple(x.shape[1:]) == (128, 15, 15) and x.dtype == self.torch.float32 and
                (self.trunk_backend == "hip16" or x.shape[0] >= 192))

    def _trunk_hip16(self, x):
        """The residual blocks on padded-row tensors [n][128][15][16] (training mode)."""
        from . import hipconv
        F = self.torch.nn.functional
        x = F.pad(x, (0, 1))                     # dense -> padded rows (pad column zero)
        p = self.p
        for i in range(1, self.n_blocks + 1):
            skip = x
            y = hipconv.conv3x3(x, p["convA%d_weight" % i], p["convA%d_bias" % i], hipconv.ROWS16)
            y = hipconv.bn_act(y, p["bnA%d_gamma" % i], p["bnA%d_beta" % i], p["bnA%d_moving_mean" % i],
                               p["bnA%d_moving_var" % i], None, True, hipconv.ROWS16, 1.0 - BN_MOMENTUM, BN_EPS)
            y = hipconv.conv3x3(y, p["convB%d_weight" % i], p["convB%d_bias" % i], hipconv.ROWS16)
            x = hipconv.bn_act(y, p["bnB%d_gamma" % i], p["bnB%d_beta" % i], p["bnB%d_moving_mean" % i],
                               p["bnB%d_moving_var" % i], skip, True, hipconv.ROWS16, 1.0 - BN_MOMENTUM, BN_EPS)
        return x[..., :15].contiguous()

    def forward(self, states, train=True):
        torch, F = self.torch, self.torch.nn.functional
        x = states
        if self.kind == "resnet":
            x = self._conv_act(x, "res_conv1", 3, train)
            hip16 = self._use_hip16(x, train)
            if hip16:
                x = self._trunk_hip16(x)
            for i in range(1, 0 if hip16 else self.n_blocks + 1):
                skip = x
                y = self._conv(x, self.p["convA%d_weight" % i], self.p["convA%d_bias" % i], 3)
                y = F.relu(self._bn(y, "bnA%d" % i, False, "_moving_mean", "_moving_var", train))
                y = self._conv(y, self.p["convB%d_weight" % i], self.p["convB%d_bias" % i], 3)
                y = self._bn(y, "bnB%d" % i, False, "_moving_mean", "_moving_var", train)
                x = F.relu(y + skip)
        else:
            for name in ("conv1", "conv2", "conv3", "conv4", "conv5", "conv_final"):
                x = self._conv_act(x, name, 3, train)
        n = x.shape[0]
        pol = self._conv_act(x, "conv3_1_1", 1, train).reshape(n, -1)
        val = self._conv_act(x, "conv3_2_1", 1, train).reshape(n, -1)
        if train and self.dropout > 0:
            keep = 1.0 - self.dropout
            pol = pol * (torch.rand(pol.shape, generator=self.gen, device=self.device, dtype=self.dtype) < keep) / keep
            val = val * (torch.rand(val.shape, generator=self.gen, device=self.device, dtype=self.dtype) < keep) / keep
        logits = pol @ self.p["fc_3_1_1_weight"].t() + self.p["fc_3_1_1_bias"]
        logp = F.log_softmax(logits, dim=1)
        v = torch.tanh(val @ self.p["fc_3_2_1_weight"].t() + self.p["fc_3_2_1_bias"])
        return logp, v

    def loss(self, states, mcts_probs, winners, train=True):
        logp, v = self.forward(states, train)
        value_loss = ((winners.reshape(-1, 1) - v) ** 2).mean()
        policy_loss = (-(logp * mcts_probs).sum(dim=1)).mean()
        entropy = (-(logp.exp() * logp).sum(dim=1)).mean()
        return value_loss + policy_loss, entropy

    # ---- one optimiser step (policy_value_net_mxnet.py:282-299) ------------------------------
    def _to(self, a, shape):
        return self.torch.as_tensor(np.asarray(a), dtype=self.dtype, device=self.device).reshape(shape)

    def train_step(self, state_batch, mcts_probs, winner_batch, learning_rate):
        torch = self.torch
        c = self.p[next(iter(self.p))].shape[1]
        hw = self.p["fc_3_1_1_bias"].shape[0]
        side = int(round(hw ** 0.5))
        states = self._to(state_batch, (-1, c, side, side))
        pis = self._to(mcts_probs, (-1, hw))
        zs = self._to(winner_batch, (-1,))
        for k in self.train_names:
            self.p[k].grad = None
        loss, entropy = self.loss(states, pis, zs, train=True)
        loss.backward()
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        lr_t = learning_rate * (1.0 - b2 ** self.t) ** 0.5 / (1.0 - b1 ** self.t)
        rescale = 1.0 / self.batch_size
        if self.conv_backend == "hip" and self.device.type == "cuda" and self.dtype == torch.float32:
            self._adam_hip(lr_t, b1, b2, eps, rescale)        # one launch over all tensors
        else:
            with torch.no_grad():
                for k in self.train_names:
                    w = self.p[k]
                    g = w.grad if w.grad is not None else torch.zeros_like(w)
                    wd = self.wd if k.endswith(("_weight", "_gamma")) else 0.0
                    g = g * rescale + wd * w
                    self.m[k].mul_(b1).add_(g, alpha=1.0 - b1)
                    self.v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
                    w.sub_(lr_t * self.m[k] / (self.v[k].sqrt() + eps))
        return float(loss.detach().cpu()), float(entropy.detach().cpu())

    def _adam_hip(self, lr_t, b1, b2, eps, rescale):
        """The same update as the loop above through apz_adam_step: a table of (w, grad, m, v, n, wd) per tensor."""
        import ctypes as C
        from . import _native, hipconv
        torch = self.torch
        tab = np.zeros(len(self.train_names), dtype=[("w", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("n", "i8"),
                                                     ("wd", "f4"), ("pad", "i4")])
        keep = []
        for i, k in enumerate(self.train_names):
            w = self.p[k]
            g = w.grad if w.grad is not None else torch.zeros_like(w)
            g = g.contiguous()
            keep.append(g)
            tab[i] = (w.data_ptr(), g.data_ptr(), self.m[k].data_ptr(), self.v[k].data_ptr(), w.numel(),
                      self.wd if k.endswith(("_weight", "_gamma")) else 0.0, 0)
        L = _native.hip()
        hnd = hipconv._engine(15, 15, self.device.index or 0)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = L.apz_adam_step(hnd, tab.ctypes.data_as(C.c_void_p), len(tab), lr_t, b1, b2, eps, rescale, stream)
        if rc < 0:
            raise RuntimeError(L.apz_last_error().decode())

    def policy_value(self, state_batch):
        """Inference-mode (moving statistics) probabilities and values, for the KL monitor."""
        torch = self.torch
        c = self.p[next(iter(self.p))].shape[1]
        hw = self.p["fc_3_1_1_bias"].shape[0]
        side = int(round(hw ** 0.5))
        with torch.no_grad():
            logp, v = self.forward(self._to(state_batch, (-1, c, side, side)), train=False)
        return logp.exp().cpu().numpy(), v.cpu().numpy()

    def get_params(self):
        return collections.OrderedDict((k, v.detach().cpu().numpy().astype(np.float32)) for k, v in self.p.items())


def policy_update(trainer, mini_batch, learn_rate=1e-3, lr_multiplier=1.0, epochs=8, kl_targ=0.02, evaluator=None):
    """train_mxnet.py:194-240: epochs x train_step with KL early stop and the adaptive LR multiplier.
    mini_batch: list of (state, mcts_prob, winner_z).  `evaluator.policy_value(states)` (the HIP
    PolicyValueNet) supplies old/new predictions when given, else the trainer's own inference
    graph.  -> (loss, entropy, kl, lr_multiplier)"""
    states = np.stack([np.ascontiguousarray(d[0]) for d in mini_batch]).astype(np.float32)
    pis = np.stack([d[1] for d in mini_batch]).astype(np.float32)
    zs = np.array([d[2] for d in mini_batch], dtype=np.float32)
    pv = (evaluator.policy_value if evaluator is not None else trainer.policy_value)
    old_probs, old_v = pv(states)
    loss = entropy = kl = 0.0
    for _ in range(epochs):
        loss, entropy = trainer.train_step(states, pis, zs, learn_rate * lr_multiplier)
        if evaluator is not None:
            evaluator.set_params(trainer.get_params())
        new_probs, new_v = pv(states)
        kl = float(np.mean(np.sum(old_probs * (np.log(old_probs + 1e-10) - np.log(new_probs + 1e-10)), axis=1)))
        if kl > kl_targ * 4:
            break
    if kl > kl_targ * 2 and lr_multiplier > 0.05:        # train_mxnet.py:215-218
        lr_multiplier /= 1.5
    elif kl < kl_targ / 2 and lr_multiplier < 20:
        lr_multiplier *= 1.5
    return loss, entropy, kl, lr_multiplier
