"""The native root sampler (libalphapig_host.so: apzh_root_sample, alphapig_amd/rootsample.py) against NumPy's own legacy
generator, bit for bit: moves, pi, and the generator state after every draw -- the calls MCTSPlayer.get_action makes
(reference mcts_alphaZero.py:13-16, :152-155, :198-201).  The reference-run fixtures pin the same thing end to end
(tests/test_host_golden.py, tests/test_selfplay_engine.py: RNG digests after every move); this file covers what those
traces cannot: every row length 1 .. 225 (np.sum's pairwise blocking), shape parameters on all three branches of the
legacy gamma sampler, temperatures, batches, and the no-noise play mode."""
import ctypes as C

import numpy as np
import pytest

from alphapig_amd import _native
from alphapig_amd._native import as_ptr
from alphapig_amd.rootsample import LegacyRngBank, sample_moves


def reference_draw(rs, row, temp, alpha, eps, with_noise):
    """The reference's own statements (mcts_alphaZero.py:13-16, :152-155, :193-201) on RandomState `rs`."""
    acts = np.flatnonzero(row >= 0)
    x = 1.0 / temp * np.log(row[acts].astype(np.int64) + 1e-10)
    probs = np.exp(x - np.max(x))
    probs /= np.sum(probs)
    pi = np.zeros(row.shape[0])
    pi[acts] = probs
    if with_noise:
        move = rs.choice(acts, p=(1.0 - eps) * probs + eps * rs.dirichlet(alpha * np.ones(len(probs))))
    else:
        move = rs.choice(acts, p=probs)
    return pi, int(move)


def random_visits(rs, hw, k, total):
    row = np.full(hw, -1, dtype=np.int32)
    acts = np.sort(rs.permutation(hw)[:k])
    row[acts] = rs.multinomial(total, rs.dirichlet(np.ones(k) * 0.5)) if total else 0
    return row


def test_np_sum_restated_for_every_length():
    L = _native.host()
    rs = np.random.RandomState(0)
    for n in list(range(1, 300)) + [511, 512, 513, 1000, 4097]:
        for scale in (1.0, 1e-8):
            a = np.ascontiguousarray(rs.rand(n) * scale + (rs.rand(n) < 0.1) * 1e3)
            assert L.apzh_np_sum(as_ptr(a, C.c_double), n) == float(np.sum(a)), n


def test_seed_is_randomstate_seed():
    bank = LegacyRngBank(3)
    for row, seed in enumerate((0, 20260000, 2 ** 32 - 1)):
        bank.seed(row, seed)
        st = np.random.RandomState(seed).get_state()
        np.testing.assert_array_equal(bank.keys[row], st[1])
        assert bank.pos[row] == st[2] == 624
    with pytest.raises(ValueError):
        bank.seed(0, 2 ** 32)


@pytest.mark.parametrize("alpha", [0.3, 0.03, 1.0, 2.5])
def test_moves_pi_and_generator_state_match_numpy(alpha):
    """Sequences of moves per game, all row lengths; alpha = 0.3 is the reference's, 1.0 / 2.5 take the exponential and the
    Marsaglia-Tsang branch (whose Gaussian cache lives in the generator state), 0.03 makes the rejection loop spin."""
    hw = 225
    gen = np.random.RandomState(7)
    lengths = list(range(1, 226, 7)) + [2, 3, 8, 9, 128, 129, 224, 225]
    bank = LegacyRngBank(len(lengths))
    refs = []
    for i in range(len(lengths)):
        bank.seed(i, 1000 + i)
        refs.append(np.random.RandomState(1000 + i))
    for rnd in range(6):
        visits = np.stack([random_visits(gen, hw, k, 399 if rnd % 2 == 0 else 37) for k in lengths])
        temps = np.where(np.arange(len(lengths)) % 3 == 0, 1.0, 0.5) if rnd >= 4 else 1.0
        pi, moves = sample_moves(bank, np.arange(len(lengths)), visits, temps, alpha=alpha, eps=0.25, n_threads=3)
        for i, k in enumerate(lengths):
            t = float(np.broadcast_to(temps, (len(lengths),))[i])
            rpi, rmove = reference_draw(refs[i], visits[i], t, alpha, 0.25, True)
            assert moves[i] == rmove, (rnd, i, k)
            np.testing.assert_array_equal(pi[i], rpi)
            st = refs[i].get_state()
            np.testing.assert_array_equal(bank.keys[i], st[1])
            assert (bank.pos[i], bank.has_gauss[i]) == (st[2], st[3]) and bank.gauss[i] == st[4], (rnd, i)


def test_play_mode_and_low_temperature():
    """is_selfplay = 0: choice(acts, p=probs) with no Dirichlet draw (mcts_alphaZero.py:204-209), at the reference's default
    temp = 1e-3, where softmax is all but one-hot; and a subset of the bank's rows in a different order."""
    hw = 64
    gen = np.random.RandomState(3)
    bank = LegacyRngBank(10)
    refs = {}
    for i in range(10):
        bank.seed(i, 50 + i)
        refs[i] = np.random.RandomState(50 + i)
    rows = np.array([7, 2, 9, 0])
    for rnd in range(5):
        visits = np.stack([random_visits(gen, hw, int(gen.randint(1, 65)), 200) for _ in rows])
        pi, moves = sample_moves(bank, rows, visits, 1e-3, with_noise=False)
        for j, r in enumerate(rows):
            rpi, rmove = reference_draw(refs[int(r)], visits[j], 1e-3, 0.3, 0.25, False)
            assert moves[j] == rmove
            np.testing.assert_array_equal(pi[j], rpi)
    for i in range(10):                                 # untouched rows stayed untouched, used rows advanced alike
        st = refs[i].get_state()
        np.testing.assert_array_equal(bank.keys[i], st[1])
        assert bank.pos[i] == st[2]
    rs = bank.random_state(7)
    assert rs.random_sample() == refs[7].random_sample()


def test_errors_and_empty_batch():
    bank = LegacyRngBank(2)
    pi, moves = sample_moves(bank, np.zeros(0, np.int64), np.zeros((0, 9), np.int32), 1.0)
    assert pi.shape == (0, 9) and moves.shape == (0,)
    with pytest.raises(ValueError):
        sample_moves(bank, [0], np.full((1, 9), -1, np.int32), 1.0)


def test_pretouch_limit_is_shared_between_the_ranks_of_a_node():
    """8 ranks starting together each read the same MemAvailable (csrc/host_tree.cpp): their limits must add up to at most
    half of it, BASELINE configs[4]'s 31 GB of arenas per rank must still be touched on a 2 TB node, and a single rank keeps
    round 3's rule (half of the memory, at most 96 GB, at least 4)."""
    f = _native.host().apzh_pretouch_limit_gb
    assert f(2000.0, 1) == 96.0 and f(64.0, 1) == 32.0 and f(6.0, 1) == 4.0
    for avail in (8.0, 16.0, 64.0, 512.0, 1500.0, 2000.0):
        assert 8 * f(avail, 8) <= 0.5 * avail + 1e-9
    assert f(2000.0, 8) >= 31.0 and f(1500.0, 8) >= 31.0
    assert f(512.0, 8) == 32.0 and f(64.0, 8) == 4.0 and f(2000.0, 0) == 96.0
