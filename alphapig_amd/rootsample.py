"""Root move sampling of MCTSPlayer.get_action (reference mcts_alphaZero.py:141-157, :187-203) for a BATCH of games,
bit for bit on NumPy's legacy generator, without a Python loop over the games.

The reference draws, per move,   probs = softmax(1/temp * log(visits + 1e-10))             (:13-16, :152-155)
                                 noise = np.random.dirichlet(0.3 * ones(len(probs)))       (:199)
                                 move  = np.random.choice(acts, p=0.75 * probs + 0.25 * noise)   (:198-200)
from the process-wide legacy MT19937.  Here every game owns one such generator state (`LegacyRngBank`: the arrays that
np.random.RandomState.get_state() returns, one row per game) and `sample_moves` makes the draws of all games that are
ready to move in ONE call of libalphapig_host.so (apzh_root_sample: standard-gamma / Dirichlet / choice restated from
NumPy's legacy algorithms, OpenMP over games).  The elementwise log / exp of the softmax stay in NumPy -- one vector call
over the concatenated rows instead of one per game -- because NumPy's float64 exp / log are its own SIMD routines, not
the C library's; everything order-dependent (np.sum's pairwise summation, the division, cumsum, searchsorted) is restated
natively and pinned against NumPy in tests/test_host_sampler.py and, through the reference's own RNG digests, in
tests/test_host_golden.py / tests/test_selfplay_engine.py.
"""
import ctypes as C

import numpy as np

from . import _native
from ._native import as_ptr


class LegacyRngBank(object):
    """One legacy np.random.RandomState per row, as plain arrays a native call can advance in place."""

    def __init__(self, n):
        self.keys = np.zeros((int(n), 624), dtype=np.uint32)
        self.pos = np.full(int(n), 624, dtype=np.int32)
        self.has_gauss = np.zeros(int(n), dtype=np.int32)
        self.gauss = np.zeros(int(n), dtype=np.float64)
        self._L = _native.host()

    def seed(self, row, seed):
        """Row `row` := np.random.RandomState(seed) (integer seed in [0, 2**32))."""
        seed = int(seed)
        if not 0 <= seed < 2 ** 32:
            raise ValueError("Seed must be between 0 and 2**32 - 1")
        self._L.apzh_mt_seed(seed, as_ptr(self.keys[row], C.c_uint32), as_ptr(self.pos[row:row + 1], C.c_int32))
        self.has_gauss[row] = 0
        self.gauss[row] = 0.0

    def set_state(self, row, rs):
        """Row `row` := the state of a np.random.RandomState (or of the np.random module)."""
        st = rs.get_state()
        self.keys[row] = st[1]
        self.pos[row] = st[2]
        self.has_gauss[row] = st[3]
        self.gauss[row] = st[4]

    def get_state(self, row):
        """The tuple np.random.RandomState.set_state() takes."""
        return ("MT19937", self.keys[row].copy(), int(self.pos[row]), int(self.has_gauss[row]), float(self.gauss[row]))

    def random_state(self, row):
        rs = np.random.RandomState(0)
        rs.set_state(self.get_state(row))
        return rs


def sample_moves(bank, rows, visits, temps, alpha=0.3, eps=0.25, with_noise=True, want_pi=True, n_threads=1):
    """visits int32 [g, HW] (-1 where the root has no child), temps float [g] or scalar; game i draws from
    bank row rows[i].  -> (pi float64 [g, HW] or None, moves int32 [g])."""
    visits = np.ascontiguousarray(visits, dtype=np.int32)
    g, hw = visits.shape
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    mask = visits >= 0
    counts = mask.sum(axis=1).astype(np.int32)
    if g == 0:
        return (np.zeros((0, hw)) if want_pi else None), np.zeros(0, np.int32)
    if (counts == 0).any():
        raise ValueError("a root without children cannot move")
    acts = np.ascontiguousarray(np.nonzero(mask)[1], dtype=np.int32)         # row-major: ascending actions per row
    starts = np.zeros(g, dtype=np.int64)
    np.cumsum(counts[:-1], out=starts[1:])
    inv_t = 1.0 / np.broadcast_to(np.asarray(temps, dtype=np.float64), (g,))
    # x = 1.0 / temp * np.log(visits + 1e-10); e = np.exp(x - np.max(x))       -- elementwise, NumPy's own routines
    x = np.repeat(inv_t, counts) * np.log(visits[mask].astype(np.int64) + 1e-10)
    e = np.exp(x - np.repeat(np.maximum.reduceat(x, starts), counts))
    pi = np.empty((g, hw), dtype=np.float64) if want_pi else None
    moves = np.empty(g, dtype=np.int32)
    # gather the generator states of the chosen rows, advance them natively, scatter them back
    k_ = np.ascontiguousarray(bank.keys[rows])
    p_ = np.ascontiguousarray(bank.pos[rows])
    h_ = np.ascontiguousarray(bank.has_gauss[rows])
    g_ = np.ascontiguousarray(bank.gauss[rows])
    L = _native.host()
    rc = L.apzh_root_sample(g, hw, as_ptr(e, C.c_double), as_ptr(acts, C.c_int32), as_ptr(counts, C.c_int32),
                            float(alpha), float(eps), 1 if with_noise else 0, as_ptr(k_, C.c_uint32),
                            as_ptr(p_, C.c_int32), as_ptr(h_, C.c_int32), as_ptr(g_, C.c_double),
                            as_ptr(pi, C.c_double) if want_pi else None, as_ptr(moves, C.c_int32), int(n_threads))
    if rc < 0:
        raise RuntimeError("apzh_root_sample: %s" % L.apzh_last_error().decode())
    bank.keys[rows] = k_
    bank.pos[rows] = p_
    bank.has_gauss[rows] = h_
    bank.gauss[rows] = g_
    return pi, moves

