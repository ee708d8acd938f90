"""Batched scheduler == N independent sequential runs (test kind 3).

The SelfPlayEngine runs G concurrent games with one leaf per game per step; every finished
episode must be bit-identical (moves, z, recorded states; pi to 1e-12) to the ORACLE's sequential
restatement of Game_AI.start_self_play driven with the same evaluator and the same per-game
seeds -- and, for the seeds the golden fixtures cover, to the reference's own episodes."""
import os
import random

import numpy as np
import pytest

from alphapig_amd.selfplay import SelfPlayEngine, episodes_to_tuples
from fakenet import fake_policy_value_batch, fake_policy_value_fn
from oracle import selfplay_ref
from oracle.board_ref import RefBoard
from oracle.mcts_ref import RefMCTSPlayer


def sequential_episode(seed, w, n_in_row, n_playout, temp):
    b = RefBoard(w, w, n_in_row)
    pl = RefMCTSPlayer(fake_policy_value_fn, c_puct=5, n_playout=n_playout, is_selfplay=1,
                       rng=np.random.RandomState(seed))
    winner, data = selfplay_ref.start_self_play(b, pl, temp=temp, pyrandom=random.Random(seed))
    return winner, b, data, pl


@pytest.mark.parametrize("pipeline", [1, 2, 3])
def test_batched_equals_sequential_8x8(pipeline):
    G, total, npl = 5, 9, 40
    eng = SelfPlayEngine(fake_policy_value_batch, 8, 8, 4, n_games=G, n_playout=npl, temp=1.0, base_seed=1000,
                         n_threads=2, pipeline=pipeline, forced_opening=False)
    eps = eng.play_games(total)
    assert [e.index for e in eps[:total]] == list(range(total))
    net_evals = 0
    for e in eps[:total]:
        # forced_opening=False: the oracle still draws random.random() first -- emulate by a
        # pyrandom whose first draw is >= 0.09 is not needed: the engine skips the draw entirely,
        # so compare against an oracle run with a stub that never takes the branch.
        b = RefBoard(8, 8, 4)
        pl = RefMCTSPlayer(fake_policy_value_fn, c_puct=5, n_playout=npl, is_selfplay=1,
                           rng=np.random.RandomState(1000 + e.index))

        class Never(object):
            def random(self):
                return 1.0
        winner, data = selfplay_ref.start_self_play(b, pl, temp=1.0, pyrandom=Never())
        assert winner == e.winner
        np.testing.assert_array_equal(np.array(b.move_list), e.moves)
        np.testing.assert_array_equal(np.array([d[2] for d in data]), e.zs)
        np.testing.assert_allclose(np.stack([d[1] for d in data]), e.pis, rtol=0, atol=1e-12)
        planes = eng.pool.codes_to_planes(e.codes, 9)
        np.testing.assert_array_equal(planes, np.stack([d[0] for d in data]).astype(np.float32))
        net_evals += pl.mcts.n_net_evals
    eng.close()


def test_batched_equals_reference_episodes_15x15(golden_dir):
    """Seeds chosen so that game 0 / 1 of the engine ARE the golden reference episodes."""
    g = np.load(os.path.join(golden_dir, "selfplay_episodes.npz"))
    for name in ("ep15_a", "ep15_forced"):
        w, n, npl, pyseed, npseed = [int(x) for x in g[name + "/meta"]]
        # engine seeds both streams with base_seed + k: reproduce by a one-game engine whose
        # streams are re-seeded to the golden pair
        eng = SelfPlayEngine(fake_policy_value_batch, w, w, n, n_games=3, n_playout=npl,
                             temp=float(g[name + "/temp"]), base_seed=0, n_threads=1, pipeline=2)
        eng._limit = 3
        for s in range(3):
            eng._start_game(s)
        # overwrite slot 1 with the golden seeds (slots 0 and 2 keep running beside it)
        eng.set_slot_rng(1, np.random.RandomState(npseed))
        eng.slots[1].pyrnd = random.Random(pyseed)
        eng.slots[1].codes, eng.slots[1].pis, eng.slots[1].movers = [], [], []
        eng.pool.reset(1, 0)
        from alphapig_amd.game_ai import draw_forced_opening, one_hot_pi
        forced = draw_forced_opening(eng.slots[1].pyrnd)
        if forced is not None:
            for mv in forced:
                eng._record(1, one_hot_pi(w * w, mv))
                eng.pool.play_move(1, mv)
        idx = eng.slots[1].index
        while not any(e.index == idx for e in eng.finished):
            eng.run_steps(32, 3)
        e = [x for x in eng.finished if x.index == idx][0]
        np.testing.assert_array_equal(e.moves, g[name + "/moves"])
        assert e.winner == int(g[name + "/winner"])
        np.testing.assert_array_equal(e.zs, g[name + "/zs"])
        np.testing.assert_array_equal(e.pis, g[name + "/pis"])       # the native sampler restates np.sum: pi bit for bit
        np.testing.assert_array_equal(eng.pool.codes_to_planes(e.codes, 9).astype(np.uint8), g[name + "/states"])
        # ... and the slot's generator ends where the reference's np.random ended (no new game was started on it)
        import hashlib
        st = eng.rng_bank.get_state(1)
        assert hashlib.sha1(st[1].tobytes() + str(st[2]).encode()).hexdigest() == str(g[name + "/digest"])
        eng.close()


def test_forced_opening_and_seed_streams():
    """Engine game k == oracle sequential game with RandomState(base+k) / Random(base+k),
    including games that take the 9 % forced-opening branch."""
    base = 4242
    forced = [k for k in range(200) if random.Random(base + k).random() < 0.09][:1]
    upto = forced[0] + 1
    eng = SelfPlayEngine(fake_policy_value_batch, 15, 15, 5, n_games=4, n_playout=12, temp=1.0,
                         base_seed=base, n_threads=2, pipeline=2)
    eps = eng.play_games(upto)
    by_index = {e.index: e for e in eps}
    for k in (0, forced[0]):
        winner, b, data, _ = sequential_episode(base + k, 15, 5, 12, 1.0)
        e = by_index[k]
        assert e.winner == winner
        np.testing.assert_array_equal(e.moves, np.array(b.move_list))
        np.testing.assert_allclose(e.pis, np.stack([d[1] for d in data]), rtol=0, atol=1e-12)
        np.testing.assert_array_equal(e.zs, np.array([d[2] for d in data]))
    assert eng.stats["forced_openings"] >= 1
    planes, pis, zs = episodes_to_tuples(eps, eng.pool)
    assert planes.shape[0] == pis.shape[0] == zs.shape[0] == sum(len(e.moves) for e in eps)
    eng.close()


def test_run_steps_counts_and_limits():
    eng = SelfPlayEngine(fake_policy_value_batch, 8, 8, 4, n_games=6, n_playout=10, temp=1.0, base_seed=7,
                         n_threads=1, pipeline=2, forced_opening=False)
    n = eng.run_steps(5)
    assert n == 30 and eng.stats["leaf_evals"] == 30       # one leaf per slot per round
    eng.close()


def test_temperature_schedule_extension():
    """BASELINE config 5's temperature schedule (a build-side extension; the reference is
    constant-temp): from ply 4 on temp=1e-3, so pi collapses onto the most visited child."""
    eng = SelfPlayEngine(fake_policy_value_batch, 8, 8, 4, n_games=4, n_playout=30, temp=1.0, base_seed=99,
                         n_threads=1, pipeline=2, forced_opening=False, temp_schedule=[(0, 1.0), (4, 1e-3)])
    eps = eng.play_games(4)
    for e in eps:
        assert (e.pis[:4].max(axis=1) < 0.999).any()            # warm plies keep a spread
        cold = e.pis[4:]                                        # cold plies: all mass on the most visited child(ren)
        assert np.all((cold < 1e-9) | (cold > cold.max(axis=1, keepdims=True) - 1e-9))
    assert eng._temp_for(0) == 1.0 and eng._temp_for(7) == 1e-3
    eng.close()
    eng2 = SelfPlayEngine(fake_policy_value_batch, 8, 8, 4, n_games=1, n_playout=5, temp_schedule=lambda ply: 0.5,
                          forced_opening=False)
    assert eng2._temp_for(3) == 0.5
    eng2.close()


def test_forced_opening_book_needs_15x15():
    import pytest
    with pytest.raises(ValueError):
        SelfPlayEngine(fake_policy_value_batch, 8, 8, 4, n_games=1, n_playout=5)


def test_laned_evaluator_routes_slots_to_lanes():
    """policy_value_net.LanedEvaluator (one evaluator handle per pipeline group): slot s runs on lane s mod k with the
    lane's own slot s div k; shapes must agree; weight changes reach every lane.  Stand-in lanes: no GPU."""
    from alphapig_amd.policy_value_net import LanedEvaluator

    class Lane(object):
        n_slots, batchsize, hw, code_stride = 3, 32, 64, 80

        def __init__(self, name):
            self.name, self.calls, self.prm = name, [], None

        def evaluate_codes(self, codes):
            self.calls.append(("eval", len(codes)))
            return self.name

        def evaluate_codes_slot(self, slot, codes):
            self.calls.append(("eval_slot", slot, len(codes)))
            return self.name

        def submit_codes_slot(self, slot, codes):
            self.calls.append(("submit", slot, len(codes)))
            return len(codes)

        def wait_slot(self, slot, n):
            self.calls.append(("wait", slot, n))
            return self.name

        def set_params(self, prm, **kw):
            self.prm = prm

        def sync(self):
            self.calls.append(("sync",))

        def close(self):
            self.calls.append(("close",))

    a, b = Lane("a"), Lane("b")
    ev = LanedEvaluator([a, b])
    assert (ev.n_slots, ev.batchsize, ev.hw, ev.code_stride) == (6, 32, 64, 80)     # two lanes x three slots each
    codes = np.zeros((5, 80), np.uint8)
    assert ev.submit_codes_slot(0, codes) == 5 and ev.wait_slot(0, 5) == "a"
    assert ev.submit_codes_slot(1, codes) == 5 and ev.wait_slot(1, 5) == "b"
    assert ev.evaluate_codes_slot(2, codes) == "a" and ev.evaluate_codes_slot(3, codes) == "b"
    assert a.calls == [("submit", 0, 5), ("wait", 0, 5), ("eval_slot", 1, 5)]
    assert b.calls == [("submit", 0, 5), ("wait", 0, 5), ("eval_slot", 1, 5)]
    ev.set_params({"w": 1})
    assert a.prm == {"w": 1} and b.prm == {"w": 1}
    ev.sync()
    ev.close()
    assert a.calls[-2:] == [("sync",), ("close",)] and b.calls[-2:] == [("sync",), ("close",)]
    c = Lane("c")
    c.batchsize = 16
    with pytest.raises(ValueError):
        LanedEvaluator([a, c])
    with pytest.raises(ValueError):
        LanedEvaluator([])


def test_gate_holds_new_rounds_and_changes_nothing_else():
    """SelfPlayEngine.gate (a threading.Event the training pipeline's trainer thread may hold clear during a policy update,
    pipeline.TrainPipeline exclusive_updates): while it is clear run_steps starts no new round; once set the run goes on and
    plays the same games as an ungated engine (game_ai.py:70-139 per game)."""
    import threading
    import time

    def make():
        return SelfPlayEngine(fake_policy_value_batch, 8, 8, 4, n_games=6, n_playout=10, temp=1.0, base_seed=7, pipeline=2,
                              forced_opening=False)
    ref = make()
    ref.run_steps(120)
    eng = make()
    eng.gate = threading.Event()                       # clear: held
    th = threading.Thread(target=eng.run_steps, args=(120,))
    th.start()
    time.sleep(0.3)
    assert th.is_alive() and eng.stats["leaf_evals"] == 0
    eng.gate.set()
    th.join(timeout=60)
    assert not th.is_alive()
    assert eng.timers["gate_s"] >= 0.25
    assert eng.stats["leaf_evals"] == ref.stats["leaf_evals"] and eng.stats["moves"] == ref.stats["moves"]
    a, b = sorted(ref.finished, key=lambda e: e.index), sorted(eng.finished, key=lambda e: e.index)
    assert len(a) == len(b) and all(np.array_equal(x.moves, y.moves) for x, y in zip(a, b))
    ref.close()
    eng.close()
