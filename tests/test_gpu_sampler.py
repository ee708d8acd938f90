"""GPU root sampler (opt-in perf mode): exact pi, statistically correct draws, determinism."""
import numpy as np
import pytest

from alphapig_amd import weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net():
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", 15, 15, 9, 1, 128, seed=0, style="bench")
    n = PolicyValueNet(15, 15, batch_size=16, n_blocks=1, n_filter=128, model_params=prm)
    yield n
    n.close()


def host_pi(row, temp):
    acts = np.flatnonzero(row >= 0)
    x = 1.0 / temp * np.log(row[acts].astype(np.float64) + 1e-10)
    p = np.exp(x - x.max())
    p /= p.sum()
    out = np.zeros(len(row))
    out[acts] = p
    return out, acts


def make_visits(rs, g, hw=225):
    v = np.full((g, hw), -1, dtype=np.int32)
    for i in range(g):
        k = int(rs.randint(1, hw + 1))
        acts = np.sort(rs.permutation(hw)[:k])
        cnt = rs.multinomial(399, rs.dirichlet(np.ones(k) * 0.3))
        v[i, acts] = cnt
    return v


def test_pi_matches_host_softmax(net):
    rs = np.random.RandomState(0)
    v = make_visits(rs, 64)
    for temp in (1.0, 0.5, 1e-3):
        pi, mv = net.sample_moves(v, temp=temp, eps=0.25, seed=1, step=2)
        for i in range(len(v)):
            want, acts = host_pi(v[i], temp)
            np.testing.assert_allclose(pi[i], want, rtol=0, atol=2e-6)
            assert mv[i] in acts
    # cold temperature without noise: always the most visited child
    pi, mv = net.sample_moves(v, temp=1e-3, eps=0.0, seed=3, step=0)
    for i in range(len(v)):
        best = np.flatnonzero(v[i] == v[i].max())
        assert mv[i] in best


def test_deterministic_in_seed_step_game(net):
    rs = np.random.RandomState(1)
    v = make_visits(rs, 32)
    a = net.sample_moves(v, seed=7, step=5)[1]
    b = net.sample_moves(v, seed=7, step=5)[1]
    c = net.sample_moves(v, seed=7, step=6)[1]
    np.testing.assert_array_equal(a, b)
    assert (a != c).any()


def test_move_frequencies_follow_pi_without_noise(net):
    row = np.full(225, -1, dtype=np.int32)
    acts = np.array([3, 50, 112, 113, 200, 224])
    row[acts] = [5, 40, 200, 100, 50, 4]
    want, _ = host_pi(row, 1.0)
    G = 4096
    v = np.tile(row, (G, 1))
    counts = np.zeros(225)
    for step in range(8):
        mv = net.sample_moves(v, temp=1.0, eps=0.0, seed=11, step=step)[1]
        counts += np.bincount(mv, minlength=225)
    n = counts.sum()
    assert counts[np.setdiff1d(np.arange(225), acts)].sum() == 0
    chi2 = ((counts[acts] - n * want[acts]) ** 2 / (n * want[acts])).sum()
    assert chi2 < 30.0            # 5 dof: p ~ 1e-5
    np.testing.assert_allclose(counts[acts] / n, want[acts], atol=0.01)


def test_dirichlet_noise_moments(net):
    """With a single dominant child and eps=1 the draw follows Dirichlet(alpha) weights:
    E[w_i] = 1/K, so every child is drawn ~ equally often; with eps=0.25 the favourite keeps
    0.75 + 0.25/K of the mass."""
    K = 8
    row = np.full(225, -1, dtype=np.int32)
    acts = np.arange(10, 10 + K) * 3
    row[acts] = 0
    row[acts[0]] = 399
    G = 4096
    v = np.tile(row, (G, 1))
    c1 = np.zeros(225)
    c25 = np.zeros(225)
    for step in range(6):
        c1 += np.bincount(net.sample_moves(v, temp=1e-3, alpha=0.3, eps=1.0, seed=5, step=step)[1], minlength=225)
        c25 += np.bincount(net.sample_moves(v, temp=1e-3, alpha=0.3, eps=0.25, seed=5, step=step)[1], minlength=225)
    f1 = c1[acts] / c1.sum()
    np.testing.assert_allclose(f1, np.full(K, 1.0 / K), atol=0.015)
    f25 = c25[acts] / c25.sum()
    assert abs(f25[0] - (0.75 + 0.25 / K)) < 0.015
    np.testing.assert_allclose(f25[1:], np.full(K - 1, 0.25 / K), atol=0.01)


def test_engine_gpu_sampler_mode_plays_legal_games():
    from alphapig_amd.policy_value_net import PolicyValueNet
    from alphapig_amd.selfplay import SelfPlayEngine
    prm = weights.init_params("simple", 8, 8, 9, seed=1, style="bench")
    net8 = PolicyValueNet(8, 8, batch_size=32, model_params=prm, net_kind="simple")
    eng = SelfPlayEngine(net8, 8, 8, 4, n_games=32, n_playout=20, temp=1.0, base_seed=3, n_threads=4, pipeline=2,
                         forced_opening=False, sampler="gpu")
    eps = eng.play_games(40)
    assert len(eps) >= 40
    for e in eps:
        assert len(set(e.moves.tolist())) == len(e.moves) and e.winner in (-1, 1, 2)
        assert np.allclose(e.pis.sum(axis=1), 1.0, atol=1e-4)
        assert set(np.unique(e.zs)) <= {-1.0, 0.0, 1.0}
    eng.close()
    net8.close()


def test_keyed_draws_follow_the_game_not_the_row(net):
    """With per-row keys (the self-play engine passes global game index << 20 | ply) a draw depends on
    (seed, key) only: permuting the rows, changing the step counter or sampling a row alone (what another rank
    or another batch composition would do) returns the same move for the same key -- and equal visit counts
    under different keys (two games, or two ranks' row 0) do not share their noise."""
    rs = np.random.RandomState(5)
    g = 64
    v = make_visits(rs, g)
    keys = ((np.arange(g, dtype=np.uint64) * 7 + 3) << np.uint64(20)) | np.uint64(11)
    pi0, mv0 = net.sample_moves(v, temp=1.0, seed=99, step=1, keys=keys)
    perm = rs.permutation(g)
    pi1, mv1 = net.sample_moves(v[perm], temp=1.0, seed=99, step=77, keys=keys[perm])
    np.testing.assert_array_equal(mv1, mv0[perm])
    np.testing.assert_array_equal(pi1, pi0[perm])
    for i in (0, 17, 63):
        _, one = net.sample_moves(v[i:i + 1], temp=1.0, seed=99, step=5, keys=keys[i:i + 1])
        assert one[0] == mv0[i]
    # identical rows, different keys: independent draws (same keys: identical)
    same = np.repeat(v[:1], 256, axis=0)
    k2 = (np.arange(256, dtype=np.uint64) << np.uint64(20))
    _, a = net.sample_moves(same, temp=1.0, seed=1, step=0, keys=k2)
    _, b = net.sample_moves(same, temp=1.0, seed=1, step=0, keys=np.zeros(256, np.uint64))
    assert len(set(a.tolist())) > 1 and len(set(b.tolist())) == 1
    # another seed: another stream
    _, c = net.sample_moves(v, temp=1.0, seed=100, step=1, keys=keys)
    assert (c != mv0).any()
