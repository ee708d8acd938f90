#!/usr/bin/env python3
"""Latency of the drop-in sequential API on the GPU box: one `PolicyValueNet.policy_value_fn(board)` call (batch 1,
policy_value_net_mxnet.py:261-280) and one `MCTSPlayer.get_action` with n_playout = 400 (mcts_alphaZero.py:187-218),
10-block net, 15x15 -- with the breakdown: GPU time per kernel class at one board (HIP events), the entry points
(planes through apz_forward_host, codes through the zero-copy slot), the host-side pieces.
Round 2: 1.0 ms per leaf, 0.43 s per move.  usage: latency_probe.py [out.json]"""
import json, sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphapig_amd import weights
from alphapig_amd.policy_value_net import PolicyValueNet
from alphapig_amd.game import Board
from alphapig_amd.mcts_alphaZero import MCTSPlayer
prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
net = PolicyValueNet(15, 15, batch_size=16, n_blocks=10, n_filter=128, model_params=prm)
b = Board(width=15, height=15, n_in_row=5); b.init_board(0)
for m in (112, 113, 97): b.do_move(m)
res = {}
def timeit(fn, reps=300, warm=20):
    for _ in range(warm): fn()
    t = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t) / reps * 1e3
res["policy_value_fn_ms"] = timeit(lambda: net.policy_value_fn(b))
state = np.ascontiguousarray(b.current_state(), dtype=np.float32)[None]
codes = b.position_codes()[None]
res["forward_planes_ms"] = timeit(lambda: net.forward_planes(state))
res["evaluate_codes_slot_ms"] = timeit(lambda: net.evaluate_codes_slot(0, codes))
res["board_current_state_ms"] = timeit(lambda: b.current_state(), 2000)
res["board_position_codes_ms"] = timeit(lambda: b.position_codes(), 2000)
net.set_profiling(1)
for _ in range(200): net.evaluate_codes_slot(0, codes)
net.sync()
res["gpu_us_per_forward_n1"] = {k: (lambda t: round(1e3 * t[0] / max(t[1], 1), 2))(net.kernel_time_ms(k)) for k in ("stem", "trunk", "head_conv", "head_fc", "forward")}
res["gpu_us_per_forward_n1"]["trunk_is_per_launch_of_20"] = True
net.set_profiling(False)
p = MCTSPlayer(net.policy_value_fn, c_puct=5, n_playout=400, is_selfplay=0)
t = time.perf_counter(); mv = p.get_action(b); res["get_action_n400_s"] = time.perf_counter() - t; res["move"] = int(mv)
t = time.perf_counter(); mv = p.get_action(b); res["get_action_n400_s_second_call"] = time.perf_counter() - t
net.close()
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
