"""BASELINE configs[4] as written -- self-play sharded over ranks FEEDING the training pipeline: every rank plays its
shard, one all-gather of the finished tuples per game batch, rank 0 runs policy_update (train_mxnet.py:194-240), the new
weights go out in one broadcast (the cross-rank form of policy_value_net_mxnet.py:295-297) and every rank's evaluator
continues with them.  CPU ranks (gloo) with a stand-in evaluator and a stand-in trainer: plumbing, not numerics -- the
real trainer runs through the same loop on the GPU (tests/test_gpu_selfplay.py, RCCL world size 1)."""
import json
import os
import random
import subprocess
import sys

import numpy as np


def _run(tmp_path, nproc, mode=None):
    here = os.path.dirname(os.path.abspath(__file__))
    port = 27500 + random.randint(0, 2000)
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(here, "_pipeline_worker.py"), str(tmp_path)]
    if mode:
        cmd.append(mode)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    return [json.load(open(tmp_path / ("pipe%d.json" % k))) for k in range(nproc)]


def test_two_rank_training_pipeline(tmp_path):
    r0, r1 = _run(tmp_path, 2)
    # every rank ran every batch and learnt rank 0's verdicts
    assert len(r0["history"]) == len(r1["history"]) == 4
    upd0 = [h for h in r0["history"] if "loss" in h]
    upd1 = [h for h in r1["history"] if "loss" in h]
    assert len(upd0) >= 2 and len(upd0) == len(upd1)
    for a, b in zip(upd0, upd1):
        assert a["batch"] == b["batch"] and a["loss"] == b["loss"] and a["kl"] == b["kl"]
    # the replay buffer and the adaptive-LR state live on rank 0; both ranks' tuples are in it (8 dihedral copies each)
    assert r1["buffer"] == 0 and r0["buffer"] > 0 and r0["buffer"] % 8 == 0
    assert r0["last_gathered"] == r1["last_gathered"] and r0["last_gathered"] > 0
    assert r0["train_steps"] >= 2 * len(upd0) - len(upd0) and r1["train_steps"] == 0      # epochs = 2, KL early stop allowed
    # one weight broadcast per update; rank 1's evaluator holds exactly rank 0's trained weights, and they moved
    assert r0["weight_broadcasts"] == r1["weight_broadcasts"] == len(upd0)
    assert r1["sets"] == len(upd0)
    np.testing.assert_array_equal(np.array(r1["w"], np.float32), np.array(r0["trainer_w"], np.float32))
    np.testing.assert_array_equal(np.array(r0["w"], np.float32), np.array(r0["trainer_w"], np.float32))
    assert r0["b"] == r1["b"] and np.abs(np.array(r0["w"])).max() > 0
    # games are sharded round-robin: rank r plays the global indices r, r + 2, ...
    assert all(i % 2 == 0 for i in r0["game_indices"]) and all(i % 2 == 1 for i in r1["game_indices"])


def test_single_process_pipeline_is_unchanged(tmp_path):
    """world size 1 through the same worker (torchrun with one process: dist stays inactive): no collectives, the update
    runs locally."""
    (r0,) = _run(tmp_path, 1)
    assert r0["world"] == 1 and r0["weight_broadcasts"] == 0 and r0["buffer"] > 0
    assert any("loss" in h for h in r0["history"])
    np.testing.assert_array_equal(np.array(r0["w"], np.float32), np.array(r0["trainer_w"], np.float32))


def test_asynchronous_schedule_nobody_waits_for_the_trainer(tmp_path):
    """BASELINE configs[4] without idle GPUs (train_mxnet.py:265-300 with the update off the self-play path): rank 0's
    policy_update runs in a trainer thread (1.2 s each here), self-play goes on in rounds of 0.15 s on BOTH ranks
    meanwhile -- rank 1's leaf-evaluation count advances inside rank 0's update intervals, rank 0's own too --, weights
    travel only when there is a new version, and at the end every rank holds the trainer's last weights."""
    r0, r1 = _run(tmp_path, 2, "async")
    assert len(r0["history"]) == len(r1["history"]) >= 3
    ups = r0["update_intervals"]
    assert r0["updates_done"] == len(ups) >= 2 and all(b - a >= 1.0 for a, b in ups)
    for r in (r0, r1):                                   # rounds that ended INSIDE an update, with self-play progress
        inside = [[(t, n) for t, n in r["round_log"] if a < t < b] for a, b in ups]
        assert any(len(x) >= 2 and x[-1][1] > x[0][1] for x in inside), (r["rank"], ups, r["round_log"])
    gaps = np.diff([t for t, _ in r1["round_log"]][:-1])
    assert gaps.max() < 1.0, gaps                        # rank 1 never sat out an update (rounds are 0.15 s; the last round drains)
    # versions only move forward, every rank installs the same ones; one broadcast per installed version, not per round
    v0 = [h["version"] for h in r0["history"]]
    assert v0 == [h["version"] for h in r1["history"]] and v0 == sorted(v0) and v0[-1] == r0["updates_done"]
    assert r0["weight_broadcasts"] == r1["weight_broadcasts"] == len(set(v for v in v0 if v > 0)) <= len(v0)
    np.testing.assert_array_equal(np.array(r1["w"], np.float32), np.array(r0["trainer_w"], np.float32))
    np.testing.assert_array_equal(np.array(r0["w"], np.float32), np.array(r0["trainer_w"], np.float32))
    # the replay buffer lives on rank 0 and holds both ranks' games; one update per game batch at most
    assert r1["buffer"] == 0 and r0["buffer"] > 0 and r0["buffer"] % 8 == 0
    collected = r0["history"][-1]["games_collected"]
    assert collected == r0["taken"] + r1["taken"] >= 120 and r1["taken"] > 0
    assert r0["updates_done"] + r0["updates_skipped"] <= collected
    assert any("win_ratio" in h for h in r0["trainer_history"])      # the arena ran in the trainer thread (check_freq = 20)
    assert all(i % 2 == 0 for i in r0["game_indices"]) and all(i % 2 == 1 for i in r1["game_indices"])
