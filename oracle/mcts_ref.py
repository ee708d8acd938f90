"""ORACLE (test infrastructure): dtype-explicit scalar restatement of the reference PUCT search.

Follows /root/reference/mcts_alphaZero.py:13-221.  The reference relies on NumPy-2
promotion rules; here every rounding step is spelled out:

  prior      P is whatever scalar the evaluator yields (np.float32 from the net).
  c_puct*P   python-scalar * np.float32 -> float32 product (weak python scalar);
             python-scalar * float64    -> float64 product.
  u          float64:  ((c_puct*P) * sqrt(float64(N_parent))) / (1 + n)     (:78-79)
  Q          three states (:30,:59):
               INT0  python int 0 (never updated)
               PYF   python float (only terminal +-1.0 / 0.0 values seen) -> float64 math
               F32   float32 ndarray shape (1,) once a net value has passed through;
                     from then on `Q += 1.0*(v-Q)/n` is all-float32: v-Q, /float32(n), Q+...
             a PYF value is rounded to float32 at the moment the first float32 value
             arrives (python float is 'weak' against a float32 array).
  score      float64(Q) + u, first maximum in child insertion order (:48-49).
"""
import numpy as np

Q_INT0, Q_PYF, Q_F32 = 0, 1, 2
f32 = np.float32


def softmax64(x):                                             # mcts_alphaZero.py:13-16
    p = np.exp(x - np.max(x))
    p /= np.sum(p)
    return p


class RefNode(object):
    __slots__ = ("parent", "children", "n", "q", "qk", "p", "cp")

    def __init__(self, parent, prior):
        self.parent = parent
        self.children = {}          # action -> RefNode, insertion ordered
        self.n = 0
        self.q = 0.0                # value held in a python float (exact for f32 too)
        self.qk = Q_INT0
        self.p = prior
        self.cp = None              # cached c_puct*P as python float

    def is_leaf(self):
        return not self.children


def _is_f32_value(v):
    return isinstance(v, np.float32) or (isinstance(v, np.ndarray) and v.dtype == np.float32)


class RefMCTS(object):
    def __init__(self, policy_value_fn, c_puct=5, n_playout=10000):
        self.root = RefNode(None, 1.0)
        self.policy = policy_value_fn
        self.c_puct = c_puct
        self.n_playout = n_playout
        self.n_net_evals = 0
        self.n_terminal_evals = 0

    # ---- TreeNode.get_value / select (:43-49, :69-80)
    def _cp(self, node):
        if node.cp is None:
            if isinstance(node.p, np.float32):
                node.cp = float(f32(self.c_puct) * node.p)       # float32 product
            else:
                node.cp = float(self.c_puct) * float(node.p)     # float64 product
        return node.cp

    def _select(self, node):
        s = float(np.sqrt(np.float64(node.n)))
        best, best_a, best_c = None, None, None
        for a, c in node.children.items():
            u = (self._cp(c) * s) / (1 + c.n)
            val = c.q + u
            if best is None or val > best:
                best, best_a, best_c = val, a, c
        return best_a, best_c

    # ---- TreeNode.update (:51-59)
    @staticmethod
    def _update(node, v, v_is_f32):
        node.n += 1
        if node.qk == Q_F32 or v_is_f32:
            q = f32(node.q)                                   # PYF -> f32 rounding happens here
            t = f32(v) - q
            t = f32(1.0) * t
            t = t / f32(node.n)
            node.q = float(q + t)
            node.qk = Q_F32
        else:
            node.q = node.q + (1.0 * (float(v) - node.q)) / node.n
            node.qk = Q_PYF

    def _backup(self, node, v, v_is_f32):                     # update_recursive :61-67
        while node is not None:
            self._update(node, v, v_is_f32)
            v = -v
            node = node.parent

    # ---- MCTS._playout (:108-139); `state` is modified in place
    def playout(self, state):
        node = self.root
        while not node.is_leaf():
            a, node = self._select(node)
            state.do_move(a)
        action_probs, leaf_value = self.policy(state)          # evaluated even if terminal
        end, winner = state.game_end()
        if not end:
            self.n_net_evals += 1
            for a, p in action_probs:
                if a not in node.children:
                    node.children[a] = RefNode(node, p)
            is32 = _is_f32_value(leaf_value)
            v = float(np.asarray(leaf_value, dtype=np.float64).reshape(-1)[0])
        else:
            self.n_terminal_evals += 1
            is32 = False
            if winner == -1:
                v = 0.0
            else:
                v = 1.0 if winner == state.get_current_player() else -1.0
        self._backup(node, -v, is32)

    def get_move_probs(self, state, temp=1e-3):               # :141-157
        for _ in range(self.n_playout):
            self.playout(state.clone())
        acts = list(self.root.children.keys())
        visits = [c.n for c in self.root.children.values()]
        probs = softmax64(1.0 / temp * np.log(np.array(visits) + 1e-10))
        return acts, probs

    def update_with_move(self, last_move):                    # :159-167
        if last_move in self.root.children:
            self.root = self.root.children[last_move]
            self.root.parent = None
        else:
            self.root = RefNode(None, 1.0)


class RefMCTSPlayer(object):
    """mcts_alphaZero.py:173-221.  `rng` is a np.random.RandomState (the reference uses
    the global legacy stream; RandomState(seed) is the same stream as np.random.seed(seed))."""

    def __init__(self, policy_value_function, c_puct=5, n_playout=2000, is_selfplay=0, rng=None):
        self.mcts = RefMCTS(policy_value_function, c_puct, n_playout)
        self.is_selfplay = is_selfplay
        self.rng = rng if rng is not None else np.random.RandomState()

    def set_player_ind(self, p):
        self.player = p

    def reset_player(self):
        self.mcts.update_with_move(-1)

    def get_action(self, board, temp=1e-3, return_prob=0):
        move_probs = np.zeros(board.width * board.height)
        if len(board.availables) == 0:
            return None
        acts, probs = self.mcts.get_move_probs(board, temp)
        move_probs[list(acts)] = probs
        if self.is_selfplay:
            noise = self.rng.dirichlet(0.3 * np.ones(len(probs)))      # drawn first (:200)
            move = self.rng.choice(acts, p=0.75 * probs + 0.25 * noise)
            self.mcts.update_with_move(move)
        else:
            move = self.rng.choice(acts, p=probs)
            self.mcts.update_with_move(-1)
        move = int(move)
        return (move, move_probs) if return_prob else move


# ----------------------------------------------------------------------------- pure MCTS
class RefPureMCTSPlayer(object):
    """Restatement of /root/reference/mcts_pure.py:13-206 (uniform priors, random rollout)."""

    def __init__(self, c_puct=5, n_playout=2000, rng=None):
        self.rng = rng if rng is not None else np.random.RandomState()
        self.c_puct = c_puct
        self.n_playout = n_playout
        self.root = RefNode(None, 1.0)
        self.last_root = None

    def set_player_ind(self, p):
        self.player = p

    def reset_player(self):
        self.root = RefNode(None, 1.0)

    def _select(self, node):
        s = float(np.sqrt(np.float64(node.n)))
        best = None
        for a, c in node.children.items():
            val = c.q + (self.c_puct * c.p * s) / (1 + c.n)     # all float64 (:83-85)
            if best is None or val > best[0]:
                best = (val, a, c)
        return best[1], best[2]

    def _rollout(self, state, limit=1000):                    # :138-157
        player = state.get_current_player()
        winner = -1
        for _ in range(limit):
            end, winner = state.game_end()
            if end:
                break
            r = self.rng.rand(len(state.availables))            # :16
            state.do_move(state.availables[int(np.argmax(r))])  # first max (:149)
        if winner == -1:
            return 0
        return 1 if winner == player else -1

    def _playout(self, state):                                # :114-136
        node = self.root
        while not node.is_leaf():
            a, node = self._select(node)
            state.do_move(a)
        end, _ = state.game_end()
        if not end:
            k = len(state.availables)
            pr = np.ones(k) / k                                  # :24 float64
            for a, p in zip(state.availables, pr):
                node.children[a] = RefNode(node, float(p))
        v = -self._rollout(state)
        while node is not None:                                  # python int/float -> float64
            node.n += 1
            node.q = node.q + (1.0 * (v - node.q)) / node.n
            v = -v
            node = node.parent

    def get_action(self, board):                              # :159-169, :196-203
        if not board.availables:
            return None
        for _ in range(self.n_playout):
            self._playout(board.clone())
        best = None
        for a, c in self.root.children.items():
            if best is None or c.n > best[0]:
                best = (c.n, a)
        self.last_root = self.root
        self.root = RefNode(None, 1.0)
        return best[1]
