#!/usr/bin/env python3
"""Capture golden vectors by RUNNING the reference's own tree / board / self-play code.

Runs only in the build container (needs /root/reference); never on the GPU box.  The
reference modules are imported read-only (PYTHONDONTWRITEBYTECODE=1, cwd outside the
tree) with inert stand-ins for their non-hot-path top-level imports (MXNet net class,
YAML config loader, SGF iterator, e-mail helper) exactly as SURVEY.md section 8(c)
describes.  Only DATA (inputs + expected outputs) is written to tests/golden/.

    PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden.py [--only NAME]

Reference entry points exercised (file:line in /root/reference):
  Board.current_state game.py:68-94, Board.do_move :117-125, has_a_winner :127-158,
  game_end :160-167; MCTSPlayer.get_action mcts_alphaZero.py:187-218 (-> MCTS
  .get_move_probs :141-157, _playout :108-139, TreeNode :19-87);
  Game_AI.start_self_play game_ai.py:70-139; TrainPipeline.get_equi_data
  train_mxnet.py:115-135; mcts_pure.MCTSPlayer.get_action mcts_pure.py:196-203,
  Game.start_play game.py:204-230.
"""
import argparse
import hashlib
import os
import random
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ALPHAPIG_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.dont_write_bytecode = True


def import_reference():
    sys.path.insert(0, REF)
    m = types.ModuleType("policy_value_net_mxnet")
    m.PolicyValueNet = object
    sys.modules["policy_value_net_mxnet"] = m
    u = types.ModuleType("utils")
    u.__path__ = []
    cl = types.ModuleType("utils.config_loader")
    cl.config_ = {"train_logging": {"version": 1}}
    sd = types.ModuleType("utils.sgf_dataIter")
    se = types.ModuleType("utils.send_email")
    u.config_loader, u.sgf_dataIter, u.send_email = cl, sd, se
    sys.modules.update({"utils": u, "utils.config_loader": cl,
                        "utils.sgf_dataIter": sd, "utils.send_email": se})
    import mcts_alphaZero, mcts_pure, game, game_ai, train_mxnet  # noqa
    return mcts_alphaZero, mcts_pure, game, game_ai, train_mxnet


def rng_digest():
    st = np.random.get_state()
    return hashlib.sha1(st[1].tobytes() + str(st[2]).encode()).hexdigest()


def qkind(q):
    """0: python int (never updated), 1: python float, 2: float32 ndarray, 3: other."""
    if isinstance(q, (int,)) and not isinstance(q, bool):
        return 0
    if isinstance(q, float) and not isinstance(q, np.floating):
        return 1
    if isinstance(q, np.ndarray) and q.dtype == np.float32:
        return 2
    if isinstance(q, np.float64):
        return 1
    return 3


def qval(q):
    return float(np.asarray(q, dtype=np.float64).reshape(-1)[0])


# ----------------------------------------------------------------------------- planes
def capture_planes(game):
    out = {}
    cases = []
    rs = np.random.RandomState(7)
    for (w, n_in_row) in ((8, 4), (15, 5), (6, 4)):
        for start_player in (0, 1):
            for nply in (0, 1, 2, 3, 4, 5, 9, 16, 23):
                if nply > w * w:
                    continue
                mv = rs.permutation(w * w)[:nply]
                cases.append((w, n_in_row, start_player, [int(x) for x in mv]))
    # edge / corner stones
    cases.append((15, 5, 0, [0, 14, 210, 224, 7, 105, 119, 217]))
    cases.append((8, 4, 1, [0, 7, 56, 63]))
    for k, (w, n, sp, mv) in enumerate(cases):
        b = game.Board(width=w, height=w, n_in_row=n)
        b.init_board(sp)
        for m in mv:
            b.do_move(m)
        st = b.current_state()
        assert st.shape == (9, w, w)
        out["c%d_meta" % k] = np.array([w, n, sp, b.current_player], dtype=np.int32)
        out["c%d_moves" % k] = np.array(mv, dtype=np.int32)
        out["c%d_planes" % k] = np.ascontiguousarray(st).astype(np.uint8)
        old = b.current_state_old()
        out["c%d_planes4" % k] = np.ascontiguousarray(old).astype(np.uint8)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "planes.npz"), **out)
    print("planes:", len(cases), "cases")


# ----------------------------------------------------------------------------- winner
def capture_winner(game):
    out = {}
    seqs = []
    rs = np.random.RandomState(11)
    # random full games to the end on several boards
    for (w, n) in ((8, 4), (15, 5), (6, 4), (5, 5), (4, 4), (9, 5)):
        for _ in range(12):
            seqs.append((w, n, 0, [int(x) for x in rs.permutation(w * w)]))
    # scripted: horizontal/vertical/both diagonals, borders, overline, row wrap (no win)
    W = 15

    def interleave(a, b):
        r = []
        for i in range(max(len(a), len(b))):
            if i < len(a):
                r.append(a[i])
            if i < len(b):
                r.append(b[i])
        return r
    filler = [200, 202, 204, 206, 208, 180, 182]
    seqs.append((W, 5, 0, interleave([0, 1, 2, 3, 4], filler)))                 # row at border
    seqs.append((W, 5, 0, interleave([10, 11, 12, 13, 14], filler)))            # row at right border
    seqs.append((W, 5, 0, interleave([11, 12, 13, 14, 15], filler)))            # wraps rows: no win
    seqs.append((W, 5, 0, interleave([0, 15, 30, 45, 60], filler)))             # column
    seqs.append((W, 5, 0, interleave([150, 165, 180 + 15, 210, 135], [1, 3, 5, 7, 9])))  # column top
    seqs.append((W, 5, 0, interleave([0, 16, 32, 48, 64], filler)))             # diag /
    seqs.append((W, 5, 0, interleave([4, 18, 32, 46, 60], filler)))             # anti-diag
    seqs.append((W, 5, 0, interleave([14, 28, 42, 56, 70], filler)))            # anti-diag from right border
    seqs.append((W, 5, 0, interleave([20, 21, 23, 24, 25, 22], [100, 130, 160, 190, 220, 221])))  # overline 6
    seqs.append((W, 5, 1, interleave([0, 1, 2, 3, 4], filler)))                 # start_player=1
    seqs.append((8, 4, 0, interleave([3, 10, 17, 24], [60, 62, 50, 52])))       # 8x8 anti-diag
    # 4x4, n=4 scripted tie (no four in a row anywhere)
    tie = [0, 1, 2, 4, 3, 5, 7, 6, 9, 8, 10, 11, 12, 13, 15, 14]
    seqs.append((4, 4, 0, tie))
    for k, (w, n, sp, mv) in enumerate(seqs):
        b = game.Board(width=w, height=w, n_in_row=n)
        b.init_board(sp)
        res = []
        used = []
        for m in mv:
            b.do_move(m)
            used.append(m)
            win, who = b.has_a_winner()
            end, winner = b.game_end()
            res.append((int(win), int(who), int(end), int(winner)))
            if end:
                break
        out["s%d_meta" % k] = np.array([w, n, sp], dtype=np.int32)
        out["s%d_moves" % k] = np.array(used, dtype=np.int32)
        out["s%d_res" % k] = np.array(res, dtype=np.int32)
    out["n_seqs"] = np.array(len(seqs))
    np.savez_compressed(os.path.join(OUT, "winner.npz"), **out)
    print("winner:", len(seqs), "sequences")


# ----------------------------------------------------------------------------- equi
def capture_equi(train_mxnet):
    out = {}
    rs = np.random.RandomState(3)
    k = 0
    for w in (3, 8, 15):
        self_ = types.SimpleNamespace(board_height=w, board_width=w)
        # index-valued inputs so outputs ARE the permutation tables
        state = np.arange(9 * w * w, dtype=np.float64).reshape(9, w, w)
        pi = np.arange(w * w, dtype=np.float64)
        ext = train_mxnet.TrainPipeline.get_equi_data(self_, [(state, pi, 1.0)])
        assert len(ext) == 8
        out["e%d_w" % k] = np.array(w)
        out["e%d_state_perm" % k] = np.stack([e[0] for e in ext]).astype(np.int32)
        out["e%d_pi_perm" % k] = np.stack([e[1] for e in ext]).astype(np.int32)
        # and a random-valued two-tuple batch
        st2 = (rs.rand(2, 9, w, w) > 0.6).astype(np.float64)
        pi2 = rs.rand(2, w * w)
        ext2 = train_mxnet.TrainPipeline.get_equi_data(
            self_, [(st2[0], pi2[0], 1.0), (st2[1], pi2[1], -1.0)])
        out["e%d_in_state" % k] = st2.astype(np.uint8)
        out["e%d_in_pi" % k] = pi2
        out["e%d_out_state" % k] = np.stack([e[0] for e in ext2]).astype(np.uint8)
        out["e%d_out_pi" % k] = np.stack([e[1] for e in ext2])
        out["e%d_out_z" % k] = np.array([e[2] for e in ext2])
        k += 1
    out["n"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "equi.npz"), **out)
    print("equi:", k, "board sizes")


# ----------------------------------------------------------------------------- search traces
def snapshot_root(root):
    acts = list(root._children.keys())
    nodes = list(root._children.values())
    return dict(
        acts=np.array(acts, dtype=np.int32),
        visits=np.array([n._n_visits for n in nodes], dtype=np.int64),
        q=np.array([qval(n._Q) for n in nodes], dtype=np.float64),
        qk=np.array([qkind(n._Q) for n in nodes], dtype=np.int8),
        p=np.array([float(n._P) for n in nodes], dtype=np.float64),
        root_n=np.array(root._n_visits, dtype=np.int64),
        root_q=np.array(qval(root._Q)),
        root_qk=np.array(qkind(root._Q), dtype=np.int8),
    )


def run_trace(mz, game, fn, w, n_in_row, n_playout, is_selfplay, temp, seed, max_moves,
              c_puct=5, pre_moves=()):
    """Repeated MCTSPlayer.get_action on one progressing board (tree reuse iff selfplay)."""
    b = game.Board(width=w, height=w, n_in_row=n_in_row)
    b.init_board()
    for m in pre_moves:
        b.do_move(m)
    player = mz.MCTSPlayer(fn, c_puct=c_puct, n_playout=n_playout, is_selfplay=is_selfplay)
    snaps = []
    orig = player.mcts.get_move_probs

    def wrapped(state, t=1e-3):
        r = orig(state, t)
        snaps.append(snapshot_root(player.mcts._root))
        return r
    player.mcts.get_move_probs = wrapped
    np.random.seed(seed)
    rec = dict(moves=[], probs=[], digests=[], ends=[])
    for _ in range(max_moves):
        mv, pr = player.get_action(b, temp=temp, return_prob=1)
        rec["moves"].append(int(mv))
        rec["probs"].append(np.array(pr, dtype=np.float64))
        rec["digests"].append(rng_digest())
        b.do_move(mv)
        end, winner = b.game_end()
        rec["ends"].append((int(end), int(winner)))
        if end:
            break
    out = {
        "meta": np.array([w, n_in_row, n_playout, is_selfplay, seed, c_puct], dtype=np.int64),
        "temp": np.array(temp, dtype=np.float64),
        "pre_moves": np.array(list(pre_moves), dtype=np.int32),
        "moves": np.array(rec["moves"], dtype=np.int32),
        "probs": np.stack(rec["probs"]),
        "ends": np.array(rec["ends"], dtype=np.int32),
        "digests": np.array(rec["digests"]),
    }
    for i, s in enumerate(snaps):
        for k2, v in s.items():
            out["m%d_%s" % (i, k2)] = v
    return out


def capture_traces(mz, game):
    from fakenet import fake_policy_value_fn, uniform_policy_value_fn
    out = {}
    specs = [
        # name, fn, w, n, playouts, selfplay, temp, seed, max_moves, pre_moves
        ("sp8_t1", fake_policy_value_fn, 8, 4, 200, 1, 1.0, 101, 64, ()),
        ("sp8_cold", fake_policy_value_fn, 8, 4, 200, 1, 1e-3, 102, 64, ()),
        ("play8_cold", fake_policy_value_fn, 8, 4, 200, 0, 1e-3, 103, 64, ()),
        ("play8_t1", fake_policy_value_fn, 8, 4, 120, 0, 1.0, 104, 64, ()),
        ("uni8", uniform_policy_value_fn, 8, 4, 150, 1, 1.0, 105, 10, ()),
        ("sp15_t1", fake_policy_value_fn, 15, 5, 400, 1, 1.0, 201, 6, ()),
        ("sp15_cold", fake_policy_value_fn, 15, 5, 400, 1, 1e-3, 202, 5, ()),
        ("sp15_small", fake_policy_value_fn, 15, 5, 60, 1, 1.0, 203, 225, ()),
        ("play15", fake_policy_value_fn, 15, 5, 100, 0, 1e-3, 204, 12, (112, 113, 97)),
        # near-terminal start: black has an open three on row 7 -> many terminal leaves
        ("sp15_tactic", fake_policy_value_fn, 15, 5, 400, 1, 1e-3, 205, 8,
         (110, 95, 111, 96, 112, 140)),
        ("sp6_full", fake_policy_value_fn, 6, 4, 80, 1, 1.0, 206, 36, ()),
    ]
    for spec in specs:
        name = spec[0]
        r = run_trace(mz, game, *spec[1:9], pre_moves=spec[9])
        for k, v in r.items():
            out[name + "/" + k] = v
        print("trace", name, "moves:", len(r["moves"]), "end:", r["ends"][-1])
    out["names"] = np.array([s[0] for s in specs])
    out["fns"] = np.array(["uniform" if s[1] is uniform_policy_value_fn else "fake" for s in specs])
    np.savez_compressed(os.path.join(OUT, "search_traces.npz"), **out)


# ----------------------------------------------------------------------------- self-play episodes
def capture_selfplay(mz, game, game_ai):
    from fakenet import fake_policy_value_fn
    out = {}
    # find python-`random` seeds that do / do not take the forced-opening branch
    forced, normal = [], []
    s = 0
    while len(forced) < 2 or len(normal) < 2:
        random.seed(s)
        (forced if random.random() < 0.09 else normal).append(s)
        s += 1
    specs = [
        ("ep15_a", 15, 5, 50, 1.0, normal[0], 301),
        ("ep15_forced", 15, 5, 40, 1.0, forced[0], 302),
        ("ep8_a", 8, 4, 100, 1.0, normal[1], 303),
        ("ep8_forced", 8, 4, 60, 1.0, forced[1], 304),
        ("ep15_400", 15, 5, 400, 1.0, normal[0], 305),
    ]
    for name, w, n, npl, temp, pyseed, npseed in specs:
        b = game.Board(width=w, height=w, n_in_row=n)
        g = game_ai.Game_AI(b)
        player = mz.MCTSPlayer(fake_policy_value_fn, c_puct=5, n_playout=npl, is_selfplay=1)
        random.seed(pyseed)
        np.random.seed(npseed)
        if w < 15 and pyseed in forced:
            # forced opening draws moves < 103 which do not exist on 8x8 -> skip that combo
            print("skip", name)
            continue
        winner, data = g.start_self_play(player, is_shown=0, temp=temp)
        data = list(data)
        states = np.stack([np.ascontiguousarray(d[0]) for d in data]).astype(np.uint8)
        pis = np.stack([d[1] for d in data]).astype(np.float64)
        zs = np.array([d[2] for d in data], dtype=np.float64)
        moves = np.array([m for m, _ in b.history], dtype=np.int32)
        out[name + "/meta"] = np.array([w, n, npl, pyseed, npseed], dtype=np.int64)
        out[name + "/temp"] = np.array(temp)
        out[name + "/winner"] = np.array(winner)
        out[name + "/moves"] = moves
        out[name + "/states"] = states
        out[name + "/pis"] = pis
        out[name + "/zs"] = zs
        out[name + "/digest"] = np.array(rng_digest())
        out[name + "/pyrandom_next"] = np.array(random.random())
        print("episode", name, "plies", len(moves), "winner", winner)
    out["names"] = np.array([s_[0] for s_ in specs if (s_[0] + "/moves") in out])
    np.savez_compressed(os.path.join(OUT, "selfplay_episodes.npz"), **out)


# ----------------------------------------------------------------------------- pure MCTS
def capture_pure(mp, game):
    out = {}
    # single get_action calls
    k = 0
    for (w, n, npl, seed, pre) in ((8, 4, 100, 401, ()), (8, 4, 100, 402, (27, 28, 35, 36)),
                                   (6, 4, 300, 403, (14, 15, 20)), (15, 5, 50, 404, (112,))):
        b = game.Board(width=w, height=w, n_in_row=n)
        b.init_board()
        for m in pre:
            b.do_move(m)
        pl = mp.MCTSPlayer(c_puct=5, n_playout=npl)
        np.random.seed(seed)
        # capture root visit counts before the tree is reset: wrap get_move
        snap = {}
        orig = pl.mcts.get_move

        def wrapped(state, _orig=orig, _pl=pl, _snap=snap):
            mv = _orig(state)
            r = _pl.mcts._root
            _snap["acts"] = np.array(list(r._children.keys()), dtype=np.int32)
            _snap["visits"] = np.array([c._n_visits for c in r._children.values()], dtype=np.int64)
            _snap["q"] = np.array([float(c._Q) for c in r._children.values()], dtype=np.float64)
            return mv
        pl.mcts.get_move = wrapped
        mv = pl.get_action(b)
        out["a%d_meta" % k] = np.array([w, n, npl, seed], dtype=np.int64)
        out["a%d_pre" % k] = np.array(pre, dtype=np.int32)
        out["a%d_move" % k] = np.array(mv)
        out["a%d_acts" % k] = snap["acts"]
        out["a%d_visits" % k] = snap["visits"]
        out["a%d_q" % k] = snap["q"]
        out["a%d_digest" % k] = np.array(rng_digest())
        k += 1
    out["n_actions"] = np.array(k)
    # BASELINE config 1: full pure-MCTS self match, 8x8, 4-in-row, n_playout=100
    for gi, seed in enumerate((411, 412)):
        b = game.Board(width=8, height=8, n_in_row=4)
        g = game.Game(b)
        p1 = mp.MCTSPlayer(c_puct=5, n_playout=100)
        p2 = mp.MCTSPlayer(c_puct=5, n_playout=100)
        np.random.seed(seed)
        winner = g.start_play(p1, p2, start_player=gi % 2, is_shown=0)
        out["g%d_meta" % gi] = np.array([8, 4, 100, seed, gi % 2], dtype=np.int64)
        out["g%d_moves" % gi] = np.array([m for m, _ in b.history], dtype=np.int32)
        out["g%d_winner" % gi] = np.array(winner)
        out["g%d_digest" % gi] = np.array(rng_digest())
        print("pure game", gi, "plies", len(b.history), "winner", winner)
    out["n_games"] = np.array(2)
    np.savez_compressed(os.path.join(OUT, "pure_mcts.npz"), **out)


# ----------------------------------------------------------------------------- SGF ingestion
SGF_FILES = {
    "0001_Blank_.sgf": "(;GM[4]FF[4]SZ[15];B[hh];W[ii];B[hi];W[ih];B[hj];W[jj];B[hk];W[kk];B[hl])\n\n",
    "0002_White_.sgf": "(;FF[4]GM[4]SZ[15];B[aa];W[oo];B[ab];W[on];B[ba];W[om];B[ca];W[ol];B[da];W[ok])\n\n",
    "0003_white_.sgf": "(;SZ[15];B[gg];W[hh])\n\n",
    "0004_blank_.sgf": "(;SZ[15];B[gg];W[hh];B[gg];W[ii])\n\n",        # repeated cell -> warning
}


def capture_sgf(game):
    """utils/sgf_dataIter.py cannot be imported whole under Python 3 (py2 print statements from
    line 134 on), so only its parsing prefix (lines 1-66) is executed.  The parser drops the
    last FOUR characters of the text it reads; the synthetic records end in "])" + two newlines
    (Python 3 text mode would fold a "\r\n" into one character)."""
    import tempfile
    src = open(os.path.join(REF, "utils", "sgf_dataIter.py"), encoding="utf-8").read()
    prefix = src[:src.index("def read_files(")]
    mod = types.ModuleType("sgf_dataIter_prefix")
    exec(compile(prefix, "sgf_dataIter.py[:read_files]", "exec"), mod.__dict__)
    game.sgf_dataIter = mod                                    # what game.py:239 calls
    out = {}
    tmp = tempfile.mkdtemp(prefix="sgf_golden_")
    names = sorted(SGF_FILES)
    for k, name in enumerate(names):
        with open(os.path.join(tmp, name), "w", newline="") as f:
            f.write(SGF_FILES[name])
        rec = mod.get_data_from_files(name, tmp + os.sep)
        out["f%d_name" % k] = np.array(name)
        out["f%d_text" % k] = np.array(SGF_FILES[name])
        out["f%d_winner" % k] = np.array(rec["winner"])
        out["f%d_seq_list" % k] = np.array(rec["seq_list"])
        out["f%d_seq_num" % k] = np.array(rec["seq_num_list"], dtype=np.int32)
        b = game.Board(width=15, height=15, n_in_row=5)
        g = game.Game(b)

        class P(object):
            def reset_player(self):
                pass
        warning, winner, data = g.start_self_play(P(), is_shown=0, sgf_home=tmp + os.sep, file_name=name)
        out["f%d_warning" % k] = np.array(warning)
        if not warning:
            data = list(data)
            out["f%d_states" % k] = np.stack([np.ascontiguousarray(d[0]) for d in data]).astype(np.uint8)
            out["f%d_pis" % k] = np.stack([d[1] for d in data])
            out["f%d_zs" % k] = np.array([d[2] for d in data])
            out["f%d_replay_winner" % k] = np.array(winner)
    out["n"] = np.array(len(names))
    np.savez_compressed(os.path.join(OUT, "sgf.npz"), **out)
    print("sgf:", len(names), "files")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    os.chdir("/tmp")
    mz, mp, game, game_ai, train_mxnet = import_reference()
    todo = {
        "planes": lambda: capture_planes(game),
        "winner": lambda: capture_winner(game),
        "equi": lambda: capture_equi(train_mxnet),
        "traces": lambda: capture_traces(mz, game),
        "selfplay": lambda: capture_selfplay(mz, game, game_ai),
        "pure": lambda: capture_pure(mp, game),
        "sgf": lambda: capture_sgf(game),
    }
    for k, f in todo.items():
        if args.only in (None, k):
            f()
    print("numpy", np.__version__, "python", sys.version.split()[0])


if __name__ == "__main__":
    main()
