"""Arena: batched evaluation matches "current net vs pure MCTS" -- the next row after self-play
(SURVEY.md section 8f rank 3).  Reference: TrainPipeline.policy_evaluate (train_mxnet.py:242-263)
driving Game.start_play (game.py:204-230) with MCTSPlayer(policy_value_fn, is_selfplay=0)
(mcts_alphaZero.py:204-209: temp 1e-3, tree reset every move) against mcts_pure.MCTSPlayer
(mcts_pure.py:196-203).

The reference plays the n_games matches one after the other; here M matches run concurrently:
the net player's searches advance in lock step (one leaf per match per step, batched through the
evaluator exactly like SelfPlayEngine), the pure-MCTS moves run in native threads.  Match i
draws everything (the net player's `np.random.choice`, the rollouts' `np.random.rand`) from its
own legacy stream RandomState(base_seed + i): it is the match the reference plays after
`np.random.seed(base_seed + i)` with `start_player = i % 2`.
"""
import collections
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .treepool import TreePool, NEED_EVAL, MOVE_READY

MatchResult = collections.namedtuple("MatchResult", "index start_player moves winner")


class Arena(object):
    def __init__(self, evaluator, board_width=15, board_height=15, n_in_row=5, n_playout=400, c_puct=5,
                 pure_mcts_playout_num=1000, base_seed=0, n_threads=0, max_concurrent=256):
        self.evaluator = evaluator
        self.w, self.h, self.n_in_row = board_width, board_height, n_in_row
        self.hw = board_width * board_height
        self.n_playout, self.c_puct = n_playout, c_puct
        self.pure_n = pure_mcts_playout_num
        self.base_seed = base_seed
        self.n_threads = n_threads
        self.max_concurrent = max_concurrent

    def _evaluate(self, pool, codes):
        if hasattr(self.evaluator, "evaluate_codes"):
            return self.evaluator.evaluate_codes(codes)
        p, v = self.evaluator(pool.codes_to_planes(codes, 9))
        return np.ascontiguousarray(p, dtype=np.float32), np.ascontiguousarray(v, dtype=np.float32).reshape(-1)

    def play(self, n_games):
        """-> list[MatchResult] in match order.  Player 1 is the net, player 2 pure MCTS."""
        out = []
        for lo in range(0, n_games, self.max_concurrent):
            out += self._play_block(list(range(lo, min(n_games, lo + self.max_concurrent))))
        return out

    def _play_block(self, indices):
        M = len(indices)
        net_pool = TreePool(self.w, self.h, self.n_in_row, n_games=M, n_playout=self.n_playout, c_puct=self.c_puct,
                            prior_is_f32=True, n_threads=self.n_threads)
        pure_pools = [TreePool(self.w, self.h, self.n_in_row, n_games=1, n_playout=self.pure_n, c_puct=5,
                               prior_is_f32=False, n_threads=1) for _ in range(M)]
        rngs = [np.random.RandomState(self.base_seed + i) for i in indices]
        start = [i % 2 for i in indices]
        active = np.ones(M, dtype=bool)
        winners = [None] * M
        for s in range(M):
            net_pool.reset(s, start[s])
            pure_pools[s].reset(0, start[s])

        def apply_move(s, mv):
            pure_pools[s].do_move(0, mv)
            ended, winner, _ = net_pool.play_move(s, mv)       # re-roots; the play-mode reset follows
            net_pool.update_with_move(s, -1)
            if ended:
                active[s] = False
                winners[s] = winner

        def pure_move(s):
            st = rngs[s].get_state()
            mv, key, pos = pure_pools[s].pure_get_move(0, st[1], st[2])
            rngs[s].set_state((st[0], key, pos) + tuple(st[3:]))
            return s, mv

        workers = ThreadPoolExecutor(max_workers=max(1, min(16, M)))
        try:
            while active.any():
                # ---- pure-MCTS side to move (player 2): native rollouts, one thread per match
                pure_turn = [s for s in range(M) if active[s] and net_pool.status(s)[0] == 2]
                if pure_turn:
                    for s, mv in workers.map(pure_move, pure_turn):
                        apply_move(s, mv)
                    continue
                # ---- net side to move (player 1) in every remaining match: batched search
                ids = np.array([s for s in range(M) if active[s]], dtype=np.int32)
                for s in ids:
                    net_pool.set_playouts_done(int(s), 0)
                while True:
                    st, codes = net_pool.advance(ids)
                    need = st == NEED_EVAL
                    if not need.any():
                        break
                    p, v = self._evaluate(net_pool, codes[need])
                    net_pool.feed(ids[need], p, v)
                visits, _ = net_pool.root_visits_dense(ids)
                for row, s in zip(visits, ids):
                    acts = np.flatnonzero(row >= 0)
                    x = 1.0 / 1e-3 * np.log(row[acts].astype(np.int64) + 1e-10)     # temp = 1e-3 (game.py:219)
                    probs = np.exp(x - np.max(x))
                    probs /= np.sum(probs)
                    apply_move(int(s), int(rngs[int(s)].choice(acts, p=probs)))
        finally:
            workers.shutdown(wait=True)
        res = []
        for k, i in enumerate(indices):
            mv, _ = net_pool.history(k)
            res.append(MatchResult(i, start[k], mv.astype(np.int32), winners[k]))
        net_pool.close()
        for pp in pure_pools:
            pp.close()
        return res


def win_ratio(results):
    """train_mxnet.py:259: (wins + 0.5 * ties) / n_games from the net's (player 1) side."""
    cnt = collections.Counter(r.winner for r in results)
    return 1.0 * (cnt[1] + 0.5 * cnt[-1]) / max(len(results), 1)


def policy_evaluate(evaluator, n_games=10, **kw):
    """Drop-in for TrainPipeline.policy_evaluate: -> win ratio of the net vs pure MCTS."""
    return win_ratio(Arena(evaluator, **kw).play(n_games))
