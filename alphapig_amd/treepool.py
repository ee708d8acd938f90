"""TreePool: NumPy-facing wrapper of the native board + PUCT tree pool (libalphapig_host.so).

One pool = G game slots.  Each slot is one reference `Board` (game.py:21-170) plus one
reference `MCTS` tree (mcts_alphaZero.py:90-170); playouts of a slot stay sequential, the
pool batches the single pending leaf of every slot.
"""
import ctypes as C

import numpy as np

from . import _native
from ._native import ApzhConfig, as_ptr

NEED_EVAL = 1
MOVE_READY = 2
Q_INT0, Q_PYF, Q_F32 = 0, 1, 2


class TreePoolError(RuntimeError):
    pass


class TreePool(object):
    def __init__(self, width, height, n_in_row, n_games=1, n_playout=400, c_puct=5,
                 prior_is_f32=True, n_threads=0):
        self.L = _native.host()
        self.width, self.height, self.n_in_row = int(width), int(height), int(n_in_row)
        self.hw = self.width * self.height
        self.n_games = int(n_games)
        self.n_playout = int(n_playout)
        self.c_puct = c_puct
        cfg = ApzhConfig(self.width, self.height, self.n_in_row, self.n_games, self.n_playout,
                         1 if prior_is_f32 else 0, int(n_threads), 0, float(c_puct))
        self._h = self.L.apzh_create(C.byref(cfg))
        if not self._h:
            raise TreePoolError(self.L.apzh_last_error().decode())
        self.code_stride = self.L.apzh_code_stride(self.height, self.width)

    def close(self):
        if getattr(self, "_h", None):
            self.L.apzh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise TreePoolError("%s (code %d)" % (self.L.apzh_last_error().decode(), rc))
        return rc

    # ---- board
    def reset(self, g, start_player=0):
        self._ck(self.L.apzh_game_reset(self._h, g, start_player))

    def set_position(self, g, moves, movers, current_player):
        mv = np.ascontiguousarray(moves, dtype=np.int16)
        mo = np.ascontiguousarray(movers, dtype=np.int8)
        self._ck(self.L.apzh_game_set_position(self._h, g, as_ptr(mv, C.c_int16), as_ptr(mo, C.c_int8),
                                               len(mv), int(current_player)))

    def do_move(self, g, move):
        self._ck(self.L.apzh_game_do_move(self._h, g, int(move)))

    def status(self, g):
        """-> (current_player, n_moves, ended, winner, last_move)"""
        out = np.zeros(5, dtype=np.int32)
        self._ck(self.L.apzh_game_status(self._h, g, as_ptr(out, C.c_int32)))
        return int(out[0]), int(out[1]), bool(out[2]), int(out[3]), int(out[4])

    def history(self, g):
        mv = np.zeros(self.hw, dtype=np.int16)
        mo = np.zeros(self.hw, dtype=np.int8)
        n = self._ck(self.L.apzh_game_history(self._h, g, as_ptr(mv, C.c_int16), as_ptr(mo, C.c_int8), self.hw))
        return mv[:n].copy(), mo[:n].copy()

    def has_a_winner(self, g):
        out = np.zeros(2, dtype=np.int32)
        self._ck(self.L.apzh_game_has_a_winner(self._h, g, as_ptr(out, C.c_int32)))
        return bool(out[0]), int(out[1])

    def codes(self, g):
        c = np.zeros(self.code_stride, dtype=np.uint8)
        self._ck(self.L.apzh_game_codes(self._h, g, as_ptr(c, C.c_uint8)))
        return c

    def codes_to_planes(self, codes, n_planes=9):
        codes = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1, self.code_stride)
        n = codes.shape[0]
        out = np.empty((n, n_planes, self.height, self.width), dtype=np.float32)
        self._ck(self.L.apzh_codes_to_planes(as_ptr(codes, C.c_uint8), n, self.height, self.width, n_planes,
                                             as_ptr(out, C.c_float)))
        return out

    # ---- search
    def advance(self, games, status_out=None, codes_out=None):
        games = np.ascontiguousarray(games, dtype=np.int32)
        n = len(games)
        st = status_out if status_out is not None else np.empty(n, dtype=np.int32)
        cd = codes_out if codes_out is not None else np.empty((n, self.code_stride), dtype=np.uint8)
        self._ck(self.L.apzh_advance(self._h, as_ptr(games, C.c_int32), n, as_ptr(st, C.c_int32),
                                     as_ptr(cd, C.c_uint8)))
        return st[:n], cd[:n]

    def feed(self, games, probs, values):
        games = np.ascontiguousarray(games, dtype=np.int32)
        probs = np.ascontiguousarray(probs, dtype=np.float32)
        values = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
        n = len(games)
        if probs.shape != (n, self.hw) or values.shape != (n,):
            raise ValueError("probs must be [n, H*W] and values [n]")
        self._ck(self.L.apzh_feed(self._h, as_ptr(games, C.c_int32), n, as_ptr(probs, C.c_float),
                                  as_ptr(values, C.c_float)))

    def feed_advance(self, games, probs, values):
        """feed(games, probs, values), then advance(games): one native call -> (status, codes)"""
        games = np.ascontiguousarray(games, dtype=np.int32)
        probs = np.ascontiguousarray(probs, dtype=np.float32)
        values = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
        n = len(games)
        if probs.shape != (n, self.hw) or values.shape != (n,):
            raise ValueError("probs must be [n, H*W] and values [n]")
        st = np.empty(n, dtype=np.int32)
        cd = np.empty((n, self.code_stride), dtype=np.uint8)
        self._ck(self.L.apzh_feed_advance(self._h, as_ptr(games, C.c_int32), n, as_ptr(probs, C.c_float),
                                          as_ptr(values, C.c_float), as_ptr(st, C.c_int32), as_ptr(cd, C.c_uint8)))
        return st, cd

    def feed_sparse(self, g, actions, priors, value, value_is_f32):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        p = np.ascontiguousarray(priors, dtype=np.float64)
        self._ck(self.L.apzh_feed_sparse(self._h, g, as_ptr(a, C.c_int32), as_ptr(p, C.c_double), len(a),
                                         float(value), 1 if value_is_f32 else 0))

    def pending_path(self, g):
        mv = np.zeros(self.hw, dtype=np.int16)
        n = self._ck(self.L.apzh_pending_path(self._h, g, as_ptr(mv, C.c_int16), self.hw))
        return mv[:n]

    def playouts_done(self, g):
        return self._ck(self.L.apzh_playouts_done(self._h, g))

    def set_playouts_done(self, g, k):
        self._ck(self.L.apzh_set_playouts_done(self._h, g, k))

    def set_n_playout(self, n):
        self.n_playout = int(n)
        self._ck(self.L.apzh_set_n_playout(self._h, int(n)))

    def set_prior_mode(self, prior_is_f32):
        self._ck(self.L.apzh_set_prior_mode(self._h, 1 if prior_is_f32 else 0))

    def node_children(self, g, node=0):
        """-> dict(acts, visits, q, qk, prior, ids, n, qkind, parent, node_q, node_prior)"""
        cap = self.hw
        acts = np.zeros(cap, dtype=np.int32)
        visits = np.zeros(cap, dtype=np.int64)
        q = np.zeros(cap, dtype=np.float64)
        qk = np.zeros(cap, dtype=np.int8)
        pr = np.zeros(cap, dtype=np.float64)
        ids = np.zeros(cap, dtype=np.int32)
        n3 = np.zeros(3, dtype=np.int64)
        nq = np.zeros(2, dtype=np.float64)
        n = self._ck(self.L.apzh_node_children(self._h, g, int(node), as_ptr(acts, C.c_int32),
                                               as_ptr(visits, C.c_int64), as_ptr(q, C.c_double),
                                               as_ptr(qk, C.c_int8), as_ptr(pr, C.c_double),
                                               as_ptr(ids, C.c_int32), cap, as_ptr(n3, C.c_int64),
                                               as_ptr(nq, C.c_double)))
        return dict(acts=acts[:n], visits=visits[:n], q=q[:n], qk=qk[:n], prior=pr[:n], ids=ids[:n],
                    n=int(n3[0]), qkind=int(n3[1]), parent=int(n3[2]), node_q=float(nq[0]),
                    node_prior=float(nq[1]))

    def root_visits_dense(self, games):
        games = np.ascontiguousarray(games, dtype=np.int32)
        n = len(games)
        v = np.empty((n, self.hw), dtype=np.int32)
        nc = np.empty(n, dtype=np.int32)
        self._ck(self.L.apzh_root_visits_dense(self._h, as_ptr(games, C.c_int32), n, as_ptr(v, C.c_int32),
                                               as_ptr(nc, C.c_int32)))
        return v, nc

    def update_with_move(self, g, move):
        self._ck(self.L.apzh_update_with_move(self._h, g, int(move)))

    def play_move(self, g, move):
        """do_move + re-root; -> (ended, winner, n_moves)"""
        out = np.zeros(3, dtype=np.int32)
        self._ck(self.L.apzh_play_move(self._h, g, int(move), as_ptr(out, C.c_int32)))
        return bool(out[0]), int(out[1]), int(out[2])

    def play_moves(self, games, moves):
        """play_move for several different games at once -> (codes_before uint8 [n, stride], movers_before int32 [n],
        ended bool [n], winner int32 [n]): the recorded (state, player) of each ply and the outcome of the move."""
        ids = np.ascontiguousarray(games, dtype=np.int32)
        mv = np.ascontiguousarray(moves, dtype=np.int32)
        n = len(ids)
        codes = np.zeros((n, self.code_stride), dtype=np.uint8)
        movers = np.zeros(n, dtype=np.int32)
        out = np.zeros((n, 3), dtype=np.int32)
        self._ck(self.L.apzh_play_moves(self._h, as_ptr(ids, C.c_int32), n, as_ptr(mv, C.c_int32), as_ptr(codes, C.c_uint8),
                                        as_ptr(movers, C.c_int32), as_ptr(out, C.c_int32)))
        return codes, movers, out[:, 0].astype(bool), out[:, 1]

    def stats(self, g):
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.L.apzh_stats(self._h, g, as_ptr(out, C.c_int64)))
        return dict(net_evals=int(out[0]), terminal_playouts=int(out[1]), live_nodes=int(out[2]),
                    peak_nodes=int(out[3]))

    def pool_info(self):
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.L.apzh_pool_info(self._h, as_ptr(out, C.c_int64)))
        return dict(arena_bytes=int(out[0]), pretouched=bool(out[1]), peak_nodes=int(out[2]), live_nodes=int(out[3]))

    def arena_bytes(self):
        return self.pool_info()["arena_bytes"]

    def pure_get_move(self, g, mt_key, mt_pos, want_children=False):
        """mcts_pure get_move on slot g with the MT19937 state (key uint32[624], pos)."""
        key = np.ascontiguousarray(mt_key, dtype=np.uint32).copy()
        pos = C.c_int32(int(mt_pos))
        cap = self.hw
        acts = np.zeros(cap, dtype=np.int32)
        visits = np.zeros(cap, dtype=np.int64)
        q = np.zeros(cap, dtype=np.float64)
        nc = C.c_int32(0)
        mv = self._ck(self.L.apzh_pure_get_move(self._h, g, as_ptr(key, C.c_uint32), C.byref(pos),
                                                as_ptr(acts, C.c_int32), as_ptr(visits, C.c_int64),
                                                as_ptr(q, C.c_double), cap, C.byref(nc)))
        if want_children:
            k = nc.value
            return mv, key, pos.value, (acts[:k], visits[:k], q[:k])
        return mv, key, pos.value
