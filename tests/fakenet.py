"""Deterministic closed-form stand-in for the policy-value net (TEST INFRASTRUCTURE).

The golden search traces in tests/golden/ were captured by driving the reference's
MCTSPlayer / Game_AI (mcts_alphaZero.py:173-221, game_ai.py:70-139) with THIS
function as `policy_value_fn`.  It uses only integer arithmetic plus one correctly
rounded IEEE division per output, so it is bit-reproducible on any host (no BLAS,
no libm), and returns exactly the types the reference evaluator returns
(policy_value_net_mxnet.py:274-280): a zip of (action:int, prob:np.float32) over the
legal moves and a float32 ndarray of shape (1,).

It is peaked (weights are squared hashes, some boosted) so that searches go deep,
reach terminal leaves and mix terminal (python-float) and net (float32) backups.
"""
import numpy as np

_K_CACHE = {}


def _tables(n_in, n_out):
    key = (n_in, n_out)
    if key not in _K_CACHE:
        i = np.arange(n_in, dtype=np.int64)[:, None]
        a = np.arange(n_out, dtype=np.int64)[None, :]
        K = (i * 131 + a * 71 + 7) % 257           # [n_in, n_out]
        bias = (3 * a[0] * a[0] + 11)               # [n_out]
        kv = (np.arange(n_in, dtype=np.int64) * 37 + 5) % 101
        _K_CACHE[key] = (K, bias, kv)
    return _K_CACHE[key]


def fake_policy_value_batch(planes):
    """planes: [n, C, H, W] array of 0/1 -> (probs float32 [n, H*W], values float32 [n])."""
    planes = np.asarray(planes)
    n, c, h, w = planes.shape
    x = np.rint(planes).astype(np.int64).reshape(n, c * h * w)
    K, bias, kv = _tables(c * h * w, h * w)
    hsh = (x @ K + bias[None, :]) % 1009                      # exact int64
    wgt = (hsh + 1) * (hsh + 1)
    wgt = np.where(hsh % 13 == 0, wgt * 40, wgt)              # a few strong moves
    tot = wgt.sum(axis=1, keepdims=True)
    probs = (wgt.astype(np.float64) / tot.astype(np.float64)).astype(np.float32)
    v = ((x @ kv) % 1601 - 800).astype(np.float64) / 1000.0
    return probs, v.astype(np.float32)


def fake_policy_value_fn(board):
    """Drop-in for PolicyValueNet.policy_value_fn (policy_value_net_mxnet.py:261-280)."""
    legal = board.availables
    st = np.ascontiguousarray(board.current_state())
    probs, v = fake_policy_value_batch(st[None])
    return zip(legal, probs[0][legal]), v[0:1]


def uniform_policy_value_fn(board):
    """Uniform float32 priors, constant small value: exercises the tie-break path."""
    legal = board.availables
    p = np.full(len(legal), 1.0 / max(len(legal), 1), dtype=np.float32)
    return zip(legal, p), np.array([0.125], dtype=np.float32)
