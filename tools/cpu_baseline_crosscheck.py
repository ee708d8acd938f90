#!/usr/bin/env python3
"""is bench.py's `cpu_baseline` (the oracle's scalar tree + oracle/net_ref.c, one batch-1 forward per
playout) representative of the REFERENCE's own CPU path?

Drives the IMPORTED reference `MCTSPlayer.get_action` (mcts_alphaZero.py:187-218 -> get_move_probs :141-157 -> _playout
:108-139, with the reference's own `Board`, game.py:21-170) and the oracle's `RefMCTSPlayer` with the SAME policy function --
`oracle.net_ref_c.CNet.policy_value_fn`, the 10-block net's vectorised CPU forward -- for a few self-play moves at
n_playout = 400 on a 15x15 board, from the same seeds, and records playouts per second for both, the share of the time
spent inside the net, and that both played the same moves.  Runs only in the build container (needs /root/reference; the
reference is imported exactly as tools/capture_golden.py does); nothing of the reference travels: the output is
profiles/r05_cpu_baseline_crosscheck.json (numbers only).

    PYTHONDONTWRITEBYTECODE=1 python tools/cpu_baseline_crosscheck.py [--moves 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.dont_write_bytecode = True


class Timed(object):
    """policy_value_fn with a stopwatch around it."""

    def __init__(self, fn):
        self.fn, self.seconds, self.calls = fn, 0.0, 0

    def __call__(self, board):
        t = time.perf_counter()
        out = self.fn(board)
        self.seconds += time.perf_counter() - t
        self.calls += 1
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--moves", type=int, default=3)
    ap.add_argument("--n-playout", type=int, default=400)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r05_cpu_baseline_crosscheck.json"))
    args = ap.parse_args()
    from capture_golden import import_reference
    mcts_alphaZero, _, game, _, _ = import_reference()
    from alphapig_amd import weights
    from oracle.board_ref import RefBoard
    from oracle.mcts_ref import RefMCTSPlayer
    from oracle.net_ref_c import CNet

    prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
    net = CNet(prm, 15, 15, 9, 128, 10, fast=True)
    res = {"what": "self-play moves at n_playout=%d, 15x15, 10-block net on one CPU core: the imported reference tree + Board "
                   "(mcts_alphaZero.py:187-218, game.py) against the oracle's (oracle/mcts_ref.py, board_ref.py), same "
                   "policy function (oracle/net_ref.c, %s), same seeds" % (args.n_playout, net.isa),
           "moves": args.moves, "n_playout": args.n_playout}
    try:
        with open("/proc/cpuinfo") as f:
            res["cpu"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    played = {}
    for who in ("reference", "oracle"):
        fn = Timed(net.policy_value_fn)
        if who == "reference":
            b = game.Board(width=15, height=15, n_in_row=5)
            b.init_board(0)
            pl = mcts_alphaZero.MCTSPlayer(fn, c_puct=5, n_playout=args.n_playout, is_selfplay=1)
            np.random.seed(123)
        else:
            b = RefBoard(15, 15, 5)
            b.init_board(0)
            pl = RefMCTSPlayer(fn, c_puct=5, n_playout=args.n_playout, is_selfplay=1, rng=np.random.RandomState(123))
        moves = []
        t0 = time.perf_counter()
        for _ in range(args.moves):
            mv, _ = pl.get_action(b, temp=1.0, return_prob=1)
            b.do_move(int(mv))
            moves.append(int(mv))
        dt = time.perf_counter() - t0
        played[who] = moves
        n = args.moves * args.n_playout
        res[who] = {"seconds": dt, "playouts_per_s": n / dt, "net_calls": fn.calls, "net_seconds": fn.seconds,
                    "net_share": fn.seconds / dt, "tree_us_per_playout": 1e6 * (dt - fn.seconds) / n, "moves": moves}
        print("%-9s %.2f s, %.1f playouts/s, net %.0f %% of the time, tree + board %.0f us per playout, moves %s"
              % (who, dt, n / dt, 100 * fn.seconds / dt, 1e6 * (dt - fn.seconds) / n, moves))
    res["same_moves"] = played["reference"] == played["oracle"]
    res["oracle_over_reference_playouts_per_s"] = res["oracle"]["playouts_per_s"] / res["reference"]["playouts_per_s"]
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("same moves: %s; oracle / reference playouts per second = %.3f -> %s"
          % (res["same_moves"], res["oracle_over_reference_playouts_per_s"], args.out))


if __name__ == "__main__":
    main()
