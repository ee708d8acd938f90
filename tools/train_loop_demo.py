#!/usr/bin/env python3
"""Closed AlphaZero loop on the GPU box, small and quick (8x8, 4-in-row): batched HIP self-play ->
8-fold augmentation -> policy_update (HIP convolutions in the training graph) -> re-folded weights
-> arena vs pure MCTS.  Prints one JSON line per evaluation; used to produce
profiles/r01_train_loop_8x8.log."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphapig_amd.pipeline import TrainPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=240)
    ap.add_argument("--check", type=int, default=60)
    args = ap.parse_args()
    conf = dict(board_width=8, board_height=8, n_in_row=4, learn_rate=2e-3, lr_multiplier=1.0, temp=1.0,
                n_playout=100, c_puct=5, buffer_size=20000, batch_size=256, epochs=4, kl_targ=0.02,
                check_freq=args.check, pure_mcts_playout_num=200, game_batch_num=args.batches, play_batch_size=4,
                concurrent_games=128, n_blocks=3, n_filter=64, eval_games=20, model_dir="/tmp/apz_models")
    tp = TrainPipeline(conf, seed=1)
    t0 = time.time()
    print(json.dumps({"initial_win_ratio_vs_pure_mcts_200": tp.policy_evaluate()}), flush=True)
    for k in range(0, args.batches, args.check):
        tp.game_batch_num = args.check
        hist = tp.run()
        last = [h for h in hist if "loss" in h][-1]
        print(json.dumps({"batches": k + args.check, "games": tp._taken, "seconds": round(time.time() - t0, 1),
                          "loss": round(last["loss"], 4), "entropy": round(last["entropy"], 4),
                          "win_ratio": hist[-1].get("win_ratio"), "lr_multiplier": round(tp.lr_multiplier, 3),
                          "leaf_evals": tp.engine.stats["leaf_evals"]}), flush=True)
    tp.close()


if __name__ == "__main__":
    main()
