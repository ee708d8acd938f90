"""Pin the CPU oracle (oracle/) against golden vectors captured from the reference itself
(tools/capture_golden.py).  Integer / index work: bit-exact.  pi: 1e-12 (exp/log ulp)."""
import hashlib
import os
import random

import numpy as np
import pytest

from oracle.board_ref import RefBoard
from oracle.mcts_ref import RefMCTSPlayer, RefPureMCTSPlayer, Q_F32, Q_INT0, Q_PYF
from oracle import selfplay_ref
from fakenet import fake_policy_value_fn, uniform_policy_value_fn


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def rs_digest(rs):
    st = rs.get_state()
    return hashlib.sha1(st[1].tobytes() + str(st[2]).encode()).hexdigest()


def test_planes(golden_dir):
    g = load(golden_dir, "planes.npz")
    for k in range(int(g["n_cases"])):
        w, n, sp, cur = [int(x) for x in g["c%d_meta" % k]]
        b = RefBoard(w, w, n)
        b.init_board(sp)
        for m in g["c%d_moves" % k]:
            b.do_move(int(m))
        assert b.current_player == cur
        st = b.current_state()
        assert st.dtype == np.float64 and st.shape == (9, w, w)
        np.testing.assert_array_equal(st.astype(np.uint8), g["c%d_planes" % k])
        np.testing.assert_array_equal(b.current_state_old().astype(np.uint8), g["c%d_planes4" % k])


def test_winner_tables(golden_dir):
    g = load(golden_dir, "winner.npz")
    n_end = 0
    for k in range(int(g["n_seqs"])):
        w, n, sp = [int(x) for x in g["s%d_meta" % k]]
        b = RefBoard(w, w, n)
        b.init_board(sp)
        for m, res in zip(g["s%d_moves" % k], g["s%d_res" % k]):
            b.do_move(int(m))
            win, who = b.has_a_winner()
            end, winner = b.game_end()
            assert (int(win), int(who), int(end), int(winner)) == tuple(int(x) for x in res)
        n_end += int(end)
    assert n_end > 50


def test_equi(golden_dir):
    g = load(golden_dir, "equi.npz")
    for k in range(int(g["n"])):
        w = int(g["e%d_w" % k])
        state = np.arange(9 * w * w, dtype=np.float64).reshape(9, w, w)
        pi = np.arange(w * w, dtype=np.float64)
        ext = selfplay_ref.equi_data([(state, pi, 1.0)], w, w)
        np.testing.assert_array_equal(np.stack([e[0] for e in ext]).astype(np.int32), g["e%d_state_perm" % k])
        np.testing.assert_array_equal(np.stack([e[1] for e in ext]).astype(np.int32), g["e%d_pi_perm" % k])
        st2 = g["e%d_in_state" % k].astype(np.float64)
        pi2 = g["e%d_in_pi" % k]
        ext2 = selfplay_ref.equi_data([(st2[0], pi2[0], 1.0), (st2[1], pi2[1], -1.0)], w, w)
        np.testing.assert_array_equal(np.stack([e[0] for e in ext2]).astype(np.uint8), g["e%d_out_state" % k])
        np.testing.assert_array_equal(np.stack([e[1] for e in ext2]), g["e%d_out_pi" % k])
        np.testing.assert_array_equal(np.array([e[2] for e in ext2]), g["e%d_out_z" % k])


def _trace_names(golden_dir):
    g = load(golden_dir, "search_traces.npz")
    return [str(x) for x in g["names"]]


def run_oracle_trace(g, name, fn):
    w, n_in_row, n_playout, is_selfplay, seed, c_puct = [int(x) for x in g[name + "/meta"]]
    temp = float(g[name + "/temp"])
    b = RefBoard(w, w, n_in_row)
    b.init_board()
    for m in g[name + "/pre_moves"]:
        b.do_move(int(m))
    rs = np.random.RandomState(seed)
    pl = RefMCTSPlayer(fn, c_puct=c_puct, n_playout=n_playout, is_selfplay=is_selfplay, rng=rs)
    moves = g[name + "/moves"]
    for i, gm in enumerate(moves):
        # run the search but look at the root BEFORE the tree is advanced
        acts, probs = pl.mcts.get_move_probs(b, temp)
        root = pl.mcts.root
        np.testing.assert_array_equal(np.array(acts), g["%s/m%d_acts" % (name, i)])
        np.testing.assert_array_equal(np.array([c.n for c in root.children.values()]),
                                      g["%s/m%d_visits" % (name, i)])
        np.testing.assert_array_equal(np.array([c.q for c in root.children.values()]),
                                      g["%s/m%d_q" % (name, i)])
        np.testing.assert_array_equal(np.array([c.qk for c in root.children.values()], dtype=np.int8),
                                      g["%s/m%d_qk" % (name, i)])
        assert root.n == int(g["%s/m%d_root_n" % (name, i)])
        assert root.q == float(g["%s/m%d_root_q" % (name, i)])
        mp = np.zeros(w * w)
        mp[list(acts)] = probs
        np.testing.assert_allclose(mp, g[name + "/probs"][i], rtol=0, atol=1e-12)
        if is_selfplay:
            noise = rs.dirichlet(0.3 * np.ones(len(probs)))
            mv = int(rs.choice(acts, p=0.75 * probs + 0.25 * noise))
            pl.mcts.update_with_move(mv)
        else:
            mv = int(rs.choice(acts, p=probs))
            pl.mcts.update_with_move(-1)
        assert mv == int(gm), (name, i)
        assert rs_digest(rs) == str(g[name + "/digests"][i])
        b.do_move(mv)
        end, winner = b.game_end()
        assert (int(end), int(winner)) == tuple(int(x) for x in g[name + "/ends"][i])


@pytest.mark.parametrize("name", ["sp8_t1", "sp8_cold", "play8_cold", "play8_t1", "uni8",
                                  "sp15_small", "play15", "sp15_tactic", "sp6_full"])
def test_search_traces(golden_dir, name):
    g = load(golden_dir, "search_traces.npz")
    assert name in _trace_names(golden_dir)
    fn = uniform_policy_value_fn if str(g["fns"][list(g["names"]).index(name)]) == "uniform" else fake_policy_value_fn
    run_oracle_trace(g, name, fn)


@pytest.mark.slow
@pytest.mark.parametrize("name", ["sp15_t1", "sp15_cold"])
def test_search_traces_400(golden_dir, name):
    g = load(golden_dir, "search_traces.npz")
    run_oracle_trace(g, name, fake_policy_value_fn)


def test_trace_q_kinds_cover_all_states(golden_dir):
    """The fixtures must exercise INT0, python-float and float32 Q states."""
    g = load(golden_dir, "search_traces.npz")
    seen = set()
    for k in g.files:
        if k.endswith("_qk"):
            seen |= set(int(x) for x in np.unique(g[k]))
    assert {Q_INT0, Q_PYF, Q_F32} <= seen


@pytest.mark.parametrize("name", ["ep15_a", "ep15_forced", "ep8_a"])
def test_selfplay_episodes(golden_dir, name):
    g = load(golden_dir, "selfplay_episodes.npz")
    if name not in [str(x) for x in g["names"]]:
        pytest.skip("episode not captured")
    w, n, npl, pyseed, npseed = [int(x) for x in g[name + "/meta"]]
    b = RefBoard(w, w, n)
    rs = np.random.RandomState(npseed)
    pl = RefMCTSPlayer(fake_policy_value_fn, c_puct=5, n_playout=npl, is_selfplay=1, rng=rs)
    winner, data = selfplay_ref.start_self_play(b, pl, temp=float(g[name + "/temp"]),
                                                pyrandom=random.Random(pyseed))
    assert winner == int(g[name + "/winner"])
    np.testing.assert_array_equal(np.array(b.move_list), g[name + "/moves"])
    np.testing.assert_array_equal(np.stack([d[0] for d in data]).astype(np.uint8), g[name + "/states"])
    np.testing.assert_allclose(np.stack([d[1] for d in data]), g[name + "/pis"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(np.array([d[2] for d in data]), g[name + "/zs"])
    assert rs_digest(rs) == str(g[name + "/digest"])


def test_pure_mcts_actions(golden_dir):
    g = load(golden_dir, "pure_mcts.npz")
    for k in range(int(g["n_actions"])):
        w, n, npl, seed = [int(x) for x in g["a%d_meta" % k]]
        b = RefBoard(w, w, n)
        b.init_board()
        for m in g["a%d_pre" % k]:
            b.do_move(int(m))
        rs = np.random.RandomState(seed)
        pl = RefPureMCTSPlayer(c_puct=5, n_playout=npl, rng=rs)
        mv = pl.get_action(b)
        root = pl.last_root
        np.testing.assert_array_equal(np.array(list(root.children.keys())), g["a%d_acts" % k])
        np.testing.assert_array_equal(np.array([c.n for c in root.children.values()]), g["a%d_visits" % k])
        np.testing.assert_array_equal(np.array([c.q for c in root.children.values()]), g["a%d_q" % k])
        assert mv == int(g["a%d_move" % k])
        assert rs_digest(rs) == str(g["a%d_digest" % k])


def test_pure_mcts_full_game_config1(golden_dir):
    """BASELINE config 1: 8x8, 4-in-row, n_playout=100 pure-MCTS self match."""
    g = load(golden_dir, "pure_mcts.npz")
    gi = 0
    w, n, npl, seed, sp = [int(x) for x in g["g%d_meta" % gi]]
    b = RefBoard(w, w, n)
    b.init_board(sp)
    rs = np.random.RandomState(seed)
    players = {1: RefPureMCTSPlayer(5, npl, rs), 2: RefPureMCTSPlayer(5, npl, rs)}
    while True:
        mv = players[b.get_current_player()].get_action(b)
        b.do_move(mv)
        end, winner = b.game_end()
        if end:
            break
    np.testing.assert_array_equal(np.array(b.move_list), g["g%d_moves" % gi])
    assert winner == int(g["g%d_winner" % gi])
    assert rs_digest(rs) == str(g["g%d_digest" % gi])
