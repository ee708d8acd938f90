"""Training step (PARITY UNPINNED), CPU side: the PyTorch comparator of the HIP trainer
(tests/torch_trainer.py) vs the NumPy float64 oracle -- loss, finite differences, Adam -- and the KL-adaptive
policy_update loop of alphapig_amd/train.py driven by it.  (The HIP trainer itself against the comparator:
tests/test_gpu_train.py.)"""
import numpy as np
import pytest

from alphapig_amd import weights
from oracle import train_ref

torch = pytest.importorskip("torch")


def tiny_problem(seed=0, n=6, w=8, blocks=1, filt=16):
    rs = np.random.RandomState(seed)
    prm = weights.init_params("resnet", w, w, 9, blocks, filt, seed=seed, style="bench")
    states = (rs.rand(n, 9, w, w) > 0.6).astype(np.float32)
    pis = rs.dirichlet(np.ones(w * w), size=n).astype(np.float32)
    zs = rs.choice([-1.0, 1.0], size=n).astype(np.float32)
    return prm, states, pis, zs


def test_loss_and_gradients_match_numpy_oracle():
    from torch_trainer import TorchTrainer
    prm, states, pis, zs = tiny_problem()
    tr = TorchTrainer(prm, "resnet", n_blocks=1, batch_size=6, device="cpu", dtype=torch.float64, dropout=0.0)
    st = torch.tensor(states, dtype=torch.float64)
    loss, ent = tr.loss(st, torch.tensor(pis, dtype=torch.float64), torch.tensor(zs, dtype=torch.float64), train=True)
    o_loss, o_ent = train_ref.train_loss(prm, states, pis, zs, 1)
    assert abs(float(loss) - o_loss) < 1e-10 and abs(float(ent) - o_ent) < 1e-10
    loss.backward()
    rs = np.random.RandomState(1)
    for name in ("res_conv1_weight", "convA1_weight", "convB1_bias", "bnA1_gamma", "bnB1_beta", "conv3_1_1_weight",
                 "fc_3_1_1_weight", "fc_3_2_1_weight", "res_conv1_beta"):
        g = tr.p[name].grad.numpy().ravel()
        for idx in rs.randint(0, g.size, size=2):
            fd = train_ref.finite_difference(prm, name, int(idx), states, pis, zs, 1)
            assert abs(g[idx] - fd) < 1e-6 + 1e-4 * abs(fd), (name, idx, g[idx], fd)
    # gamma of fix_gamma layers takes no gradient (frozen at 1)
    assert tr.p["res_conv1_gamma"].grad is None or float(tr.p["res_conv1_gamma"].grad.abs().max()) == 0.0


def test_adam_update_rule():
    from torch_trainer import TorchTrainer
    prm, states, pis, zs = tiny_problem(seed=3)
    tr = TorchTrainer(prm, "resnet", n_blocks=1, batch_size=128, device="cpu", dtype=torch.float64, dropout=0.0)
    # expected: two steps of the NumPy rule driven by the trainer's own gradients
    names = ("convA1_weight", "convA1_bias", "bnA1_gamma", "fc_3_1_1_bias")
    w = {k: prm[k].astype(np.float64) for k in names}
    m = {k: np.zeros_like(w[k]) for k in names}
    v = {k: np.zeros_like(w[k]) for k in names}
    for t in (1, 2):
        st = torch.tensor(states, dtype=torch.float64)
        for k in tr.train_names:
            tr.p[k].grad = None
        loss, _ = tr.loss(st, torch.tensor(pis, dtype=torch.float64), torch.tensor(zs, dtype=torch.float64))
        loss.backward()
        grads = {k: tr.p[k].grad.numpy().copy() for k in names}
        tr.train_step(states, pis, zs, 2e-3)
        for k in names:
            wd = 1e-4 if k.endswith(("_weight", "_gamma")) else 0.0
            w[k], m[k], v[k] = train_ref.adam_step(w[k], grads[k], m[k], v[k], t, 2e-3, wd, 128)
            np.testing.assert_allclose(tr.p[k].detach().numpy(), w[k], rtol=0, atol=1e-12)
    # moving statistics moved towards the batch statistics, trainable tensors only in train_names
    assert not np.allclose(tr.get_params()["bnA1_moving_mean"], prm["bnA1_moving_mean"])
    assert "bnA1_moving_mean" not in tr.train_names
    # gammas of the fix_gamma layers are not optimised and read 1, as MXNet leaves them (it rewrites them on forward)
    out = tr.get_params()
    for k in ("res_conv1_gamma", "conv3_1_1_gamma", "conv3_2_1_gamma"):
        assert k not in tr.train_names and np.all(out[k] == 1.0)
    assert "bnA1_gamma" in tr.train_names and "bnB1_gamma" in tr.train_names


def test_policy_update_reduces_loss_and_adapts_lr():
    from alphapig_amd.train import policy_update
    from torch_trainer import TorchTrainer
    prm, states, pis, zs = tiny_problem(seed=5, n=32)
    tr = TorchTrainer(prm, "resnet", n_blocks=1, batch_size=32, device="cpu", dropout=0.5, seed=1)
    batch = [(states[i], pis[i], zs[i]) for i in range(32)]
    first = None
    mult = 1.0
    mons = []
    for it in range(6):
        mon = {}
        old_v = tr.policy_value(states)[1]
        loss, ent, kl, mult = policy_update(tr, batch, learn_rate=5e-3, lr_multiplier=mult, epochs=3, kl_targ=0.02, monitors=mon)
        first = loss if first is None else first
        assert np.isfinite(loss) and np.isfinite(ent) and kl >= -1e-6
        # the value head's monitors (train_mxnet.py:222-227), restated here in the reference's own words
        want_old = 1 - np.var(np.array(zs) - old_v.flatten()) / np.var(np.array(zs))
        want_new = 1 - np.var(np.array(zs) - tr.policy_value(states)[1].flatten()) / np.var(np.array(zs))
        assert abs(mon["explained_var_old"] - want_old) < 1e-6 and abs(mon["explained_var_new"] - want_new) < 1e-6
        mons.append(mon)
    assert loss < first
    assert 0.05 / 1.5 <= mult <= 20 * 1.5
    assert mons[-1]["explained_var_new"] > mons[0]["explained_var_old"]      # the value head learns the 32 outcomes


def test_product_trainer_has_no_cpu_path():
    """alphapig_amd.train.HipTrainer runs on the HIP kernels only: without a GPU it refuses to construct."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from alphapig_amd.train import HipTrainer
    prm, _, _, _ = tiny_problem()
    with pytest.raises(RuntimeError):
        HipTrainer(prm, "resnet", n_blocks=1, batch_size=6)


def test_replay_buffer_is_the_reference_deque_with_random_access():
    """train_mxnet.py:59 / :196: deque(maxlen) + random.sample.  Same contents, order and sampled mini-batches."""
    import collections
    import random
    from alphapig_amd.pipeline import ReplayBuffer
    buf, dq = ReplayBuffer(37), collections.deque(maxlen=37)
    for step in range(9):
        chunk = [(step, k) for k in range(step * 3 + 1)]
        buf.extend(chunk)
        dq.extend(chunk)
        assert len(buf) == len(dq) and list(buf) == list(dq) and buf[0] == dq[0] and buf[-1] == dq[-1]
        if len(dq) >= 5:
            a, b = random.Random(step), random.Random(step)
            assert buf.sample(a, 5) == b.sample(list(dq), 5) and a.random() == b.random()
    with pytest.raises(IndexError):
        buf[37]
