#!/usr/bin/env python3
"""The reference's training configuration (conf/train_config.yaml: 15x15, 5-in-row, n_playout 400, c_puct 5, batch 128,
8 epochs per update, learn_rate 4e-4, kl_targ 0.02, one policy update per finished self-play game, pure-MCTS opponent
with 1000 playouts) on ONE MI355X: 1024 concurrent self-play games feed the game queue, every update re-folds the
evaluator device to device.  A slice of `--batches` game batches with timing, then one arena evaluation.
Prints JSON lines (profiles/r02_train_loop_15x15.log)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphapig_amd.pipeline import TrainPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=600)
    ap.add_argument("--report", type=int, default=100)
    ap.add_argument("--eval-games", type=int, default=10)
    ap.add_argument("--pure-playouts", type=int, default=1000)
    args = ap.parse_args()
    conf = dict(board_width=15, board_height=15, n_in_row=5, learn_rate=4e-4, lr_multiplier=1.0, temp=1.0,
                n_playout=400, c_puct=5, buffer_size=2198800, batch_size=128, epochs=8, kl_targ=0.02,
                check_freq=10 ** 9, pure_mcts_playout_num=args.pure_playouts, game_batch_num=args.report,
                play_batch_size=1, concurrent_games=1024, n_blocks=10, n_filter=128, eval_games=args.eval_games,
                model_dir="/tmp/apz_models_15")
    tp = TrainPipeline(conf, seed=1)
    t0 = time.time()
    upd_s = 0.0
    orig_update = tp.policy_update

    def timed_update():
        nonlocal upd_s
        t = time.time()
        r = orig_update()
        upd_s += time.time() - t
        return r

    tp.policy_update = timed_update
    done = 0
    while done < args.batches:
        g0, e0, u0, tt = tp._taken, tp.engine.stats["leaf_evals"], upd_s, time.time()
        hist = tp.run()
        done += args.report
        dt = time.time() - tt
        last = [h for h in hist if "loss" in h]
        rec = {"batches": done, "games_taken": tp._taken, "seconds": round(time.time() - t0, 1),
               "games_per_s_this_window": round((tp._taken - g0) / dt, 2),
               "leaf_evals_per_s_this_window": round((tp.engine.stats["leaf_evals"] - e0) / dt),
               "policy_update_ms": round(1e3 * (upd_s - u0) / max(1, len([h for h in hist[-args.report:] if "loss" in h])), 1),
               "update_share_of_wall": round((upd_s - u0) / dt, 3), "buffer": len(tp.data_buffer),
               "lr_multiplier": round(tp.lr_multiplier, 3)}
        if last:
            rec.update(loss=round(last[-1]["loss"], 4), entropy=round(last[-1]["entropy"], 4), kl=round(last[-1]["kl"], 5))
        print(json.dumps(rec), flush=True)
    t = time.time()
    wr = tp.policy_evaluate()
    print(json.dumps({"arena_games": args.eval_games, "pure_mcts_playouts": args.pure_playouts, "win_ratio": wr,
                      "arena_seconds": round(time.time() - t, 1)}), flush=True)
    tp.close()


if __name__ == "__main__":
    main()
