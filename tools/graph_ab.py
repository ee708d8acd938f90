#!/usr/bin/env python3
"""A/B of apz_submit_codes' HIP-graph replay (apz_set_forward_graphs) on the GPU box: BASELINE config 2 (64 concurrent 8x8
games on the 6-conv net, 32-board forwards) and the one-board latency path of the 10-block net."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphapig_amd import weights  # noqa: E402
from alphapig_amd.game import Board  # noqa: E402
from alphapig_amd.mcts_alphaZero import MCTSPlayer  # noqa: E402
from alphapig_amd.policy_value_net import PolicyValueNet  # noqa: E402
from alphapig_amd.selfplay import SelfPlayEngine  # noqa: E402


def config2(graphs, steps=6000):
    prm = weights.init_params("simple", 8, 8, 9, 10, 128, seed=0, style="bench")
    net = PolicyValueNet(8, 8, batch_size=32, n_blocks=10, n_filter=128, model_params=prm, net_kind="simple")
    net._ck(net.L.apz_set_forward_graphs(net._h, int(graphs)))
    eng = SelfPlayEngine(net, 8, 8, 4, n_games=64, n_playout=200, temp=1.0, base_seed=77, pipeline=2, forced_opening=False)
    eng.run_steps(50)
    net.sync()
    l0 = eng.stats["leaf_evals"]
    t = time.perf_counter()
    eng.run_steps(steps)
    net.sync()
    dt = time.perf_counter() - t
    r = {"graphs": graphs, "leaf_evals_per_s": (eng.stats["leaf_evals"] - l0) / dt, "ms_per_step": 1e3 * dt / steps,
         "host_tree_s": eng.timers["host_s"], "evaluator_s": eng.timers["eval_s"]}
    eng.close()
    net.close()
    return r


def latency(graphs):
    prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
    net = PolicyValueNet(15, 15, batch_size=16, n_blocks=10, n_filter=128, model_params=prm)
    net._ck(net.L.apz_set_forward_graphs(net._h, int(graphs)))
    b = Board(width=15, height=15, n_in_row=5)
    b.init_board(0)
    for m in (112, 113, 97, 127):
        b.do_move(m)
    for _ in range(30):
        net.policy_value_fn(b)
    t = time.perf_counter()
    for _ in range(300):
        net.policy_value_fn(b)
    fn_ms = 1e3 * (time.perf_counter() - t) / 300
    pl = MCTSPlayer(net.policy_value_fn, c_puct=5, n_playout=400, is_selfplay=0)
    pl.get_action(b)
    t = time.perf_counter()
    for _ in range(3):
        pl.reset_player()
        pl.get_action(b)
    ga = (time.perf_counter() - t) / 3
    net.close()
    return {"graphs": graphs, "policy_value_fn_ms": fn_ms, "get_action_n_playout_400_s": ga}


def main():
    out = {"config2": [config2(g) for g in (False, True, False, True)], "latency": [latency(g) for g in (False, True, False, True)]}
    for k, v in out.items():
        for r in v:
            print(k, json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
