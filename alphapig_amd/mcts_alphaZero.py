"""MCTS / MCTSPlayer: drop-in for the reference's AlphaZero-style search
(reference mcts_alphaZero.py:90-221), backed by the native tree pool.

`MCTSPlayer(policy_value_function, c_puct=5, n_playout=2000, is_selfplay=0)` accepts ANY
callable with the reference contract `fn(board) -> (iterable[(action, prob)], value)`
(policy_value_net_mxnet.py:261-280) and produces, for the same board + RNG state, the same
move, the same visit counts and the same pi as the reference: select / expand / backup run
in libalphapig_host.so with the reference's float32/float64 rounding.

Differences visible to a caller: terminal leaves are not sent to the evaluator (the
reference evaluates them and discards the result, :124-136), and `TreeNode` objects are
read-only views.  For throughput use `alphapig_amd.selfplay.SelfPlayEngine`, which batches
one leaf per concurrent game through the HIP evaluator.
"""
import copy

import numpy as np

from .treepool import TreePool, NEED_EVAL, Q_F32, Q_INT0


def softmax(x):
    e = np.exp(x - np.max(x))
    return e / np.sum(e)


class TreeNode(object):
    """Read-only view of one native tree node (attribute names of mcts_alphaZero.py:19-87)."""

    def __init__(self, pool, node_id, action=None):
        self._pool, self._id, self.action = pool, node_id, action

    def _info(self):
        return self._pool.node_children(0, self._id)

    @property
    def _n_visits(self):
        return self._info()["n"]

    @property
    def _Q(self):
        d = self._info()
        if d["qkind"] == Q_INT0:
            return 0
        return np.array([d["node_q"]], dtype=np.float32) if d["qkind"] == Q_F32 else d["node_q"]

    @property
    def _P(self):
        return self._info()["node_prior"]

    @property
    def _children(self):
        d = self._info()
        return {int(a): TreeNode(self._pool, int(i), int(a)) for a, i in zip(d["acts"], d["ids"])}

    @property
    def _parent(self):
        par = self._info()["parent"]
        return None if par < 0 else TreeNode(self._pool, par)

    def is_leaf(self):
        return len(self._info()["acts"]) == 0

    def is_root(self):
        return self._info()["parent"] < 0


class MCTS(object):
    def __init__(self, policy_value_fn, c_puct=5, n_playout=10000):
        self._policy = policy_value_fn
        self._c_puct = c_puct
        self._n_playout = n_playout
        self._pool = None
        self._pending_reset = False

    # the pool is sized from the first board seen
    def _bind(self, state):
        if self._pool is None or (self._pool.width, self._pool.height, self._pool.n_in_row) != \
                (state.width, state.height, state.n_in_row):
            self._pool = TreePool(state.width, state.height, state.n_in_row, n_games=1,
                                  n_playout=self._n_playout, c_puct=self._c_puct, prior_is_f32=True,
                                  n_threads=1)
            self._pool.reset(0)
        return self._pool

    def _sync_position(self, state):
        pool = self._bind(state)
        hist = getattr(state, "history", None)
        if hist is None:
            raise TypeError("board must expose .history [(move, player), ...]")
        pool.set_position(0, [m for m, _ in hist], [p for _, p in hist], state.get_current_player())
        return pool

    def _evaluate_pending(self, pool, state):
        leaf = copy.deepcopy(state)
        for mv in pool.pending_path(0):
            leaf.do_move(int(mv))
        action_probs, leaf_value = self._policy(leaf)
        pairs = list(action_probs)
        acts = [int(a) for a, _ in pairs]
        if pairs:
            p0 = pairs[0][1]
            is32 = isinstance(p0, np.float32)
            pool.set_prior_mode(is32)
        priors = [float(p) for _, p in pairs]
        v_is32 = isinstance(leaf_value, np.float32) or \
            (isinstance(leaf_value, np.ndarray) and leaf_value.dtype == np.float32)
        v = float(np.asarray(leaf_value, dtype=np.float64).reshape(-1)[0])
        pool.feed_sparse(0, acts, priors, v, v_is32)

    def _run(self, pool, state, n_playout):
        pool.set_n_playout(n_playout)
        pool.set_playouts_done(0, 0)
        g = np.zeros(1, dtype=np.int32)
        while True:
            st, _ = pool.advance(g)
            if st[0] != NEED_EVAL:
                return
            self._evaluate_pending(pool, state)

    def _playout(self, state):
        """One playout from the current root on (a copy of) `state` (mcts_alphaZero.py:108-139)."""
        self._run(self._sync_position(state), state, 1)

    def get_move_probs(self, state, temp=1e-3):
        """n_playout sequential playouts, then visit-count softmax (mcts_alphaZero.py:141-157)."""
        pool = self._sync_position(state)
        self._run(pool, state, self._n_playout)
        root = pool.node_children(0, 0)
        acts = tuple(int(a) for a in root["acts"])
        visits = tuple(int(v) for v in root["visits"])
        act_probs = softmax(1.0 / temp * np.log(np.array(visits) + 1e-10))
        return acts, act_probs

    def update_with_move(self, last_move):
        """Keep the subtree under `last_move`, else start a fresh tree (:159-167)."""
        if self._pool is not None:
            self._pool.update_with_move(0, int(last_move))

    @property
    def _root(self):
        return TreeNode(self._pool, 0) if self._pool is not None else None

    def __str__(self):
        return "MCTS"


class MCTSPlayer(object):
    """AI player (mcts_alphaZero.py:173-221).  `rng` defaults to the global legacy NumPy
    stream the reference draws from, so `np.random.seed(s)` reproduces its moves."""

    def __init__(self, policy_value_function, c_puct=5, n_playout=2000, is_selfplay=0, rng=None):
        self.mcts = MCTS(policy_value_function, c_puct, n_playout)
        self._is_selfplay = is_selfplay
        self._rng = rng if rng is not None else np.random

    def set_player_ind(self, p):
        self.player = p

    def reset_player(self):
        self.mcts.update_with_move(-1)

    def get_action(self, board, temp=1e-3, return_prob=0):
        move_probs = np.zeros(board.width * board.height)
        if len(board.availables) == 0:
            print("WARNING: the board is full")
            return None
        acts, probs = self.mcts.get_move_probs(board, temp)
        move_probs[list(acts)] = probs
        if self._is_selfplay:
            # Dirichlet(0.3) mixed into the sampling distribution only
            noisy = 0.75 * probs + 0.25 * self._rng.dirichlet(0.3 * np.ones(len(probs)))
            move = self._rng.choice(acts, p=noisy)
            self.mcts.update_with_move(move)
        else:
            move = self._rng.choice(acts, p=probs)
            self.mcts.update_with_move(-1)
        return (move, move_probs) if return_prob else move

    def __str__(self):
        return "MCTS {}".format(self.player)
