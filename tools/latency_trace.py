#!/usr/bin/env python3
"""200 single-board forwards through the zero-copy slot -- the program to put behind `rocprofv3 --kernel-trace` when the
question is where one board's 0.4 ms go (kernel durations vs the gaps between dependent launches)."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphapig_amd import weights
from alphapig_amd.policy_value_net import PolicyValueNet
from alphapig_amd.game import Board
prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
net = PolicyValueNet(15, 15, batch_size=16, n_blocks=10, n_filter=128, model_params=prm)
b = Board(width=15, height=15, n_in_row=5); b.init_board(0)
for m in (112, 113, 97): b.do_move(m)
codes = b.position_codes()[None]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200): net.evaluate_codes_slot(0, codes)
net.close()
