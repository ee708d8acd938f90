#!/usr/bin/env python3
"""Latency of the drop-in sequential API on the GPU box: one `PolicyValueNet.policy_value_fn(board)` call (batch 1,
policy_value_net_mxnet.py:261-280) and one `MCTSPlayer.get_action` with n_playout = 400 (mcts_alphaZero.py:187-218),
10-block net, 15x15.  Round 2: 1.0 ms per leaf, 0.43 s per move."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from alphapig_amd import weights
from alphapig_amd.policy_value_net import PolicyValueNet
from alphapig_amd.game import Board
from alphapig_amd.mcts_alphaZero import MCTSPlayer
prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
net = PolicyValueNet(15, 15, batch_size=16, n_blocks=10, n_filter=128, model_params=prm)
b = Board(width=15, height=15, n_in_row=5); b.init_board(0)
for m in (112, 113, 97): b.do_move(m)
for _ in range(20): net.policy_value_fn(b)
t = time.perf_counter()
for _ in range(300): net.policy_value_fn(b)
dt = (time.perf_counter() - t) / 300
print("policy_value_fn latency %.3f ms" % (dt * 1e3))
p = MCTSPlayer(net.policy_value_fn, c_puct=5, n_playout=400, is_selfplay=0)
t = time.perf_counter(); mv = p.get_action(b); print("get_action(n_playout=400): %.2f s, move %d" % (time.perf_counter() - t, mv))
net.close()
