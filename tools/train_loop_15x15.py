#!/usr/bin/env python3
"""The reference's training configuration (conf/train_config.yaml: 15x15, 5-in-row, n_playout 400, c_puct 5, batch 128,
8 epochs per update, learn_rate 4e-4, kl_targ 0.02, one policy update per finished self-play game, pure-MCTS opponent
with 1000 playouts) on ONE MI355X, asynchronous schedule (alphapig_amd/pipeline.py): 1024 concurrent self-play games
step on while the trainer thread updates; every published version is installed device to device.
Prints one JSON line per reporting window (profiles/r04_train_loop_15x15.log); --lock-step runs round 3's loop;
--max-update-share caps the share of the wall clock the trainer may be busy."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphapig_amd.pipeline import TrainPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1500, help="game batches (= games: play_batch_size 1) to collect")
    ap.add_argument("--report-s", type=float, default=20.0)
    ap.add_argument("--max-update-share", type=float, default=1.0)
    ap.add_argument("--lock-step", action="store_true")
    ap.add_argument("--exclusive", action="store_true", help="no new self-play round while the trainer is inside a policy update (default: interleaved)")
    ap.add_argument("--eval-games", type=int, default=10)
    ap.add_argument("--pure-playouts", type=int, default=1000)
    args = ap.parse_args()
    conf = dict(board_width=15, board_height=15, n_in_row=5, learn_rate=4e-4, lr_multiplier=1.0, temp=1.0,
                n_playout=400, c_puct=5, buffer_size=2198800, batch_size=128, epochs=8, kl_targ=0.02,
                check_freq=10 ** 9, pure_mcts_playout_num=args.pure_playouts, game_batch_num=args.games,
                play_batch_size=1, concurrent_games=1024, n_blocks=10, n_filter=128, eval_games=args.eval_games,
                model_dir="/tmp/apz_models_15", async_update=not args.lock_step, round_seconds=0.25,
                max_update_share=args.max_update_share, exclusive_updates=args.exclusive)
    tp = TrainPipeline(conf, seed=1)
    t0 = time.time()
    if args.lock_step:
        hist = tp.run()
        print(json.dumps({"mode": "lock step", "batches": len(hist), "seconds": round(time.time() - t0, 1),
                          "leaf_evals_per_s": round(tp.engine.stats["leaf_evals"] / (time.time() - t0))}), flush=True)
        tp.close()
        return
    # report from a side thread: the pipeline's own loop is not interrupted
    import threading
    stop = threading.Event()

    def reporter():
        last_t, last_leaf, last_busy, last_upd, last_games = t0, 0, 0.0, 0, 0
        while not stop.wait(args.report_s):
            now = time.time()
            leaf = tp.engine.stats["leaf_evals"]
            iv = list(tp.update_intervals)
            busy = sum(b - a for a, b in iv)
            games = tp._taken
            rec = {"seconds": round(now - t0, 1), "games_taken": games,
                   "games_per_s_this_window": round((games - last_games) / (now - last_t), 2),
                   "leaf_evals_per_s_this_window": round((leaf - last_leaf) / (now - last_t)),
                   "updates_done": tp.updates_done, "updates_skipped": tp.updates_skipped,
                   "policy_update_ms": round(1e3 * (busy - last_busy) / max(1, len(iv) - last_upd), 1),
                   "update_share_of_wall": round((busy - last_busy) / (now - last_t), 3), "buffer": len(tp.data_buffer),
                   "weights_version": tp.weights_version, "lr_multiplier": round(tp.lr_multiplier, 3)}
            ups = [h for h in getattr(tp, "trainer_history", []) if "loss" in h]
            if ups:
                rec.update(loss=round(ups[-1]["loss"], 4), entropy=round(ups[-1]["entropy"], 4), kl=round(ups[-1]["kl"], 5))
            print(json.dumps(rec), flush=True)
            last_t, last_leaf, last_busy, last_upd, last_games = now, leaf, busy, len(iv), games

    th = threading.Thread(target=reporter, daemon=True)
    th.start()
    tp.run()
    stop.set()
    dt = time.time() - t0
    print(json.dumps({"mode": "asynchronous", "max_update_share": args.max_update_share, "seconds": round(dt, 1),
                      "games": tp._taken, "updates_done": tp.updates_done, "updates_skipped": tp.updates_skipped,
                      "leaf_evals_per_s_whole_run": round(tp.engine.stats["leaf_evals"] / dt),
                      "update_share_of_wall_whole_run": round(sum(b - a for a, b in tp.update_intervals) / dt, 3),
                      "exclusive_updates": tp.exclusive_updates,
                      "self_play_held_s": round(tp.engine.timers.get("gate_s", 0.0), 2)}), flush=True)
    tp.close()


if __name__ == "__main__":
    main()
