#!/usr/bin/env python3
"""BASELINE.md section 3 table (B1/B2/B4): BASELINE.json configs 1-3 on this machine.

  config 1  8x8, 4-in-row, n_playout=100, pure-MCTS rollouts, CPU only: the sequential oracle
            (python restatement of mcts_pure.py) vs the native host library, same seeds
  config 2  8x8, 4-in-row, n_playout=200, simple net, 64 concurrent games on the GPU
  config 3  15x15, 5-in-row, n_playout=400, 10-block residual net, 1024 concurrent games (short slice)
  config 5  per-GPU slice of the competition-strength setting: 15x15, n_playout=1600, c_puct=5, Dirichlet 0.3,
            temperature schedule (temp 1.0 for the first 30 plies, then 0.1: a build-side extension, the reference
            plays at one constant temp -- SURVEY F5), 1024 concurrent games (configs 4 / 5 shard this over 8 GPUs)
Prints one JSON object.  Run on the GPU box:  python tests/config_table.py
(Lives under tests/: config 1 times the CPU oracle, which only tests, smoke() and bench.py's cpu_baseline may import.)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from alphapig_amd import mcts_pure, weights  # noqa: E402
from alphapig_amd.game import Board, Game  # noqa: E402


def config1(n_games=4):
    from oracle.board_ref import RefBoard
    from oracle.mcts_ref import RefPureMCTSPlayer
    out = {}
    # native
    t = time.perf_counter()
    plies = 0
    for g in range(n_games):
        b = Board(width=8, height=8, n_in_row=4)
        np.random.seed(9000 + g)
        Game(b).start_play(mcts_pure.MCTSPlayer(5, 100), mcts_pure.MCTSPlayer(5, 100), start_player=g % 2, is_shown=0)
        plies += len(b.history)
    dt = time.perf_counter() - t
    out["native_host_library"] = {"games_per_s": n_games / dt, "playouts_per_s": plies * 100 / dt, "cores": 1}
    # sequential python oracle (same algorithm as the reference's mcts_pure.py)
    t = time.perf_counter()
    plies2 = 0
    for g in range(2):
        b = RefBoard(8, 8, 4)
        b.init_board(g % 2)
        rs = np.random.RandomState(9000 + g)
        players = {1: RefPureMCTSPlayer(5, 100, rs), 2: RefPureMCTSPlayer(5, 100, rs)}
        while True:
            b.do_move(players[b.get_current_player()].get_action(b))
            if b.game_end()[0]:
                break
        plies2 += len(b.move_list)
    dt2 = time.perf_counter() - t
    out["python_oracle"] = {"games_per_s": 2 / dt2, "playouts_per_s": plies2 * 100 / dt2, "cores": 1}
    return out


def gpu_config(kind, w, nrow, npl, G, steps, n_blocks=10, temp_schedule=None, lanes=1, pipeline=2, trunk_arith="auto"):
    """lanes > 1: one evaluator handle (own HIP stream, own buffers) per pipeline group -- policy_value_net.LanedEvaluator."""
    from alphapig_amd.policy_value_net import LanedEvaluator, PolicyValueNet
    from alphapig_amd.selfplay import SelfPlayEngine
    prm = weights.init_params(kind, w, w, 9, n_blocks, 128, seed=0, style="bench")
    net = PolicyValueNet(w, w, batch_size=max(16, G // pipeline), n_blocks=n_blocks, n_filter=128, model_params=prm,
                         net_kind=kind, trunk_arith=trunk_arith)
    if lanes > 1:
        net = LanedEvaluator.like(net, lanes)
    eng = SelfPlayEngine(net, w, w, nrow, n_games=G, n_playout=npl, temp=1.0, base_seed=77, pipeline=pipeline,
                         forced_opening=(w == 15), temp_schedule=temp_schedule)
    info0 = eng.pool.pool_info()
    eng.run_steps(30)
    net.sync()
    l0, m0, g0, p0 = eng.stats["leaf_evals"], eng.stats["moves"], eng.stats["games"], eng.stats["plies"]
    t = time.perf_counter()
    eng.run_steps(steps)
    net.sync()
    dt = time.perf_counter() - t
    res = {"concurrent_games": G, "steps": steps, "leaf_evals_per_s": (eng.stats["leaf_evals"] - l0) / dt,
           "moves_per_s": (eng.stats["moves"] - m0) / dt, "games_finished": eng.stats["games"] - g0,
           "ms_per_step": 1e3 * dt / steps, "host_tree_s": eng.timers["host_s"], "evaluator_s": eng.timers["eval_s"],
           "tree_arena_gb": info0["arena_bytes"] / 1e9, "tree_arena_pretouched": info0["pretouched"],
           "peak_tree_nodes": eng.pool.pool_info()["peak_nodes"], "evaluator_lanes": lanes, "pipeline_groups": pipeline,
           "trunk_arith": net.trunk_arith, "trunk_overflows": net.trunk_overflows()}
    if eng.stats["games"] - g0 > 0:
        res["mean_plies_finished"] = (eng.stats["plies"] - p0) / (eng.stats["games"] - g0)
        res["games_per_s_finished"] = (eng.stats["games"] - g0) / dt
    eng.close()
    net.close()
    return res


def config2_roofline(n=32, trunk_arith="auto"):
    """The six convolutions of the simple net at the batch BASELINE config 2 launches (32 boards = half of the 64
    concurrent games): conv8_kernel's time per launch (HIP events over 200 back-to-back launches, apz_conv3x3_bench)
    against the fp32 matrix pipe.  The dominant kernel of the configuration is the 256 -> 256 layer."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, 9, 10, 128, seed=0, style="bench")
    net = PolicyValueNet(8, 8, batch_size=n, model_params=prm, net_kind="simple", trunk_arith=trunk_arith)
    planes = (np.random.RandomState(3).rand(n, 9, 8, 8) < 0.2).astype(np.float32)
    net.forward_planes(planes)
    chans = [9, 64, 64, 128, 128, 256, 256]
    layers = []
    for li in range(6):
        ms = net.conv_bench(li, n, iters=200, warmup=20)
        cin, cout = chans[li], chans[li + 1]
        flops = 2.0 * n * cin * cout * 9 * 64
        executed = n * (cout // 16) * ((cin + 3) // 4) * 9 * 4 * 2048.0      # MFMAs issued x 2048 flop
        layers.append({"layer": "%d->%d" % (cin, cout), "us_per_launch": 1e3 * ms, "algorithmic_gflop": flops / 1e9,
                       "executed_tflops": executed / ms / 1e9, "frac_of_fp32_mfma_peak": executed / ms / 1e9 / 157.3,
                       "workgroups": n * (cout // 16)})
    net.close()
    dom = layers[-1]
    return {"boards_per_launch": n, "layers": layers,
            "roofline": {"kernel": "conv8_kernel<false,false>, 256 -> 256 at 8x8 (csrc/conv8_small.h): work item = board x 16 output "
                                   "channels, contraction split over the four waves", "bound": "mfma",
                         "achieved": dom["executed_tflops"], "peak": 157.3, "unit": "TFLOP/s", "frac": dom["frac_of_fp32_mfma_peak"],
                         "us_per_launch": dom["us_per_launch"], "traffic": None,
                         "note": "a 32-board launch is 512 workgroups = two per CU, one round: ~9 us of the launch are fixed costs "
                                 "(launch, first input / weight round trip, reduction + store), the MFMA time alone is 16 us"}}


def main():
    out = {"config1_pure_mcts_8x8_n100_cpu": config1()}
    out["config2_simple_net_8x8_n200_64games"] = gpu_config("simple", 8, 4, 200, 64, 6000, lanes=2)
    out["config2_simple_net_8x8_n200_64games_one_lane"] = gpu_config("simple", 8, 4, 200, 64, 6000, lanes=1)
    out["config2_simple_net_8x8_n200_64games"]["kernels"] = config2_roofline(32)
    out["config3_resnet10_15x15_n400_1024games"] = gpu_config("resnet", 15, 5, 400, 1024, 400)
    out["config5_slice_resnet10_15x15_n1600_1024games"] = gpu_config("resnet", 15, 5, 1600, 1024, 2000,
                                                                     temp_schedule=[(0, 1.0), (30, 0.1)])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
