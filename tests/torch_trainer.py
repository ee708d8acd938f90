"""PyTorch restatement of the reference's training graph -- TEST COMPARATOR ONLY (tests/ and tools/train_bench.py's
comparison leg).  The product trainer is alphapig_amd/train.py's HipTrainer, which runs every operator on this
repository's HIP kernels; this file is what it is checked against, and what tests/test_train.py checks against the
NumPy float64 oracle (oracle/train_ref.py) on the CPU.

Same graph, loss and optimiser as the reference (policy_value_net_mxnet.py:41-102, :173-212, :282-299):
training-mode BatchNorm (eps 1e-3, momentum 0.9, gamma frozen at 1 on the fix_gamma layers), Dropout(0.5) on both
flattened head inputs, loss = mean((z - v)^2) + mean(-sum(pi log p)), MXNet Adam with rescale_grad = 1/batch_size
and wd on *_weight / *_gamma.  `mask_fn(which, step, shape)` (optional) supplies the dropout keep masks, so that a
test can hand it the masks of the HIP dropout kernel; `relu_masks` (optional, {layer: bool tensor}) replaces each
ReLU by a multiplication with the given mask, so that a float64 run takes the same ReLU decisions as the float32 run
it is compared with (activations within rounding distance of zero otherwise make the comparison a lottery).
"""
import collections

import numpy as np

BN_EPS = 1e-3
BN_MOMENTUM = 0.9


class TorchTrainer(object):
    def __init__(self, params, net_kind="resnet", n_blocks=10, batch_size=512, wd=1e-4, device="cpu", dtype=None,
                 dropout=0.5, seed=0, mask_fn=None, relu_masks=None):
        import torch
        self.torch = torch
        self.kind, self.n_blocks = net_kind, n_blocks
        self.batch_size, self.wd, self.dropout = batch_size, wd, dropout
        self.mask_fn = mask_fn
        self.relu_masks = relu_masks
        self.device = torch.device(device)
        self.dtype = dtype or torch.float32
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.p = collections.OrderedDict()
        for k, v in params.items():
            self.p[k] = torch.tensor(np.asarray(v), dtype=self.dtype, device=self.device)
        self.stat_names = [k for k in self.p if k.endswith(("_mean", "_var", "_moving_mean", "_moving_var"))]
        self.fixed_gamma_names = [k for k in self.p if k.endswith("_gamma") and not k.startswith(("bnA", "bnB"))]
        for k in self.fixed_gamma_names:
            self.p[k].fill_(1.0)
        self.train_names = [k for k in self.p if k not in self.stat_names and k not in self.fixed_gamma_names]
        for k in self.train_names:
            self.p[k].requires_grad_(True)
        self.m = {k: torch.zeros_like(self.p[k]) for k in self.train_names}
        self.v = {k: torch.zeros_like(self.p[k]) for k in self.train_names}
        self.t = 0

    def _bn(self, x, name, fix_gamma, mean_n, var_n, train):
        F = self.torch.nn.functional
        gamma = self.torch.ones_like(self.p[name + "_beta"]) if fix_gamma else self.p[name + "_gamma"]
        return F.batch_norm(x, self.p[name + mean_n], self.p[name + var_n], gamma, self.p[name + "_beta"], training=train,
                            momentum=1.0 - BN_MOMENTUM, eps=BN_EPS)

    def _relu(self, x, name):
        if self.relu_masks is not None:
            return x * self.relu_masks[name].to(device=x.device, dtype=x.dtype)
        return self.torch.nn.functional.relu(x)

    def _conv_act(self, x, name, k, train):
        F = self.torch.nn.functional
        y = F.conv2d(x, self.p[name + "_weight"], self.p[name + "_bias"], padding=k // 2)
        return self._relu(self._bn(y, name, True, "_mean", "_var", train), name)

    def _mask(self, which, shape):
        if self.mask_fn is not None:
            return self.mask_fn(which, self.t, shape).to(device=self.device, dtype=self.dtype)
        keep = 1.0 - self.dropout
        return (self.torch.rand(shape, generator=self.gen, device=self.device, dtype=self.dtype) < keep).to(self.dtype)

    def forward(self, states, train=True):
        torch, F = self.torch, self.torch.nn.functional
        x = states
        if self.kind == "resnet":
            x = self._conv_act(x, "res_conv1", 3, train)
            for i in range(1, self.n_blocks + 1):
                skip = x
                y = F.conv2d(x, self.p["convA%d_weight" % i], self.p["convA%d_bias" % i], padding=1)
                y = self._relu(self._bn(y, "bnA%d" % i, False, "_moving_mean", "_moving_var", train), "bnA%d" % i)
                y = F.conv2d(y, self.p["convB%d_weight" % i], self.p["convB%d_bias" % i], padding=1)
                y = self._bn(y, "bnB%d" % i, False, "_moving_mean", "_moving_var", train)
                x = self._relu(y + skip, "block%d" % i)
        else:
            for name in ("conv1", "conv2", "conv3", "conv4", "conv5", "conv_final"):
                x = self._conv_act(x, name, 3, train)
        n = x.shape[0]
        pol = self._conv_act(x, "conv3_1_1", 1, train).reshape(n, -1)
        val = self._conv_act(x, "conv3_2_1", 1, train).reshape(n, -1)
        if train and self.dropout > 0:
            keep = 1.0 - self.dropout
            pol = pol * self._mask(0, pol.shape) / keep
            val = val * self._mask(1, val.shape) / keep
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

    def grads(self):
        return collections.OrderedDict((k, self.p[k].grad.detach().cpu().numpy()) for k in self.train_names
                                       if self.p[k].grad is not None)

    def policy_value(self, state_batch):
        torch = self.torch
        c = self.p[next(iter(self.p))].shape[1]
        hw = self.p["fc_3_1_1_bias"].shape[0]
        side = int(round(hw ** 0.5))
        with torch.no_grad():
            logp, v = self.forward(self._to(state_batch, (-1, c, side, side)), train=False)
        return logp.exp().cpu().numpy(), v.cpu().numpy()

    def get_params(self):
        return collections.OrderedDict((k, v.detach().cpu().numpy().astype(np.float32)) for k, v in self.p.items())
