"""Pure-rollout MCTS player: drop-in for reference mcts_pure.py (BASELINE config 1, CPU only).

Uniform priors, leaf value from a uniformly random rollout, most-visited move, tree reset
every move (mcts_pure.py:20-25, :114-169, :196-203).  The playouts and rollouts run in
libalphapig_host.so and draw from the SAME legacy MT19937 stream as the reference's
`np.random.rand` calls: the global NumPy state is read before and written back after each
move, so `np.random.seed(s)` reproduces the reference's games move for move.
"""
import numpy as np

from .treepool import TreePool


def policy_value_fn(board):
    """Uniform priors and zero score (mcts_pure.py:20-25)."""
    k = len(board.availables)
    return zip(board.availables, np.ones(k) / k), 0


def rollout_policy_fn(board):
    return zip(board.availables, np.random.rand(len(board.availables)))


class MCTSPlayer(object):
    def __init__(self, c_puct=5, n_playout=2000, rng=None):
        self._c_puct, self._n_playout = c_puct, n_playout
        self._rng = rng if rng is not None else np.random
        self._pool = None
        self.last_children = None

    def set_player_ind(self, p):
        self.player = p

    def reset_player(self):
        pass                                      # the tree never outlives a move

    def _bind(self, board):
        if self._pool is None or (self._pool.width, self._pool.height, self._pool.n_in_row) != \
                (board.width, board.height, board.n_in_row):
            self._pool = TreePool(board.width, board.height, board.n_in_row, n_games=1,
                                  n_playout=self._n_playout, c_puct=self._c_puct, prior_is_f32=False,
                                  n_threads=1)
        return self._pool

    def get_action(self, board):
        if len(board.availables) == 0:
            print("WARNING: the board is full")
            return None
        pool = self._bind(board)
        pool.set_position(0, [m for m, _ in board.history], [p for _, p in board.history],
                          board.get_current_player())
        st = self._rng.get_state()
        move, key, pos, kids = pool.pure_get_move(0, st[1], st[2], want_children=True)
        self._rng.set_state((st[0], key, pos) + tuple(st[3:]))
        self.last_children = kids
        return move

    def __str__(self):
        return "MCTS {}".format(self.player)
