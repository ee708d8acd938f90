"""Worker for tests/test_pipeline_dist.py: the multi-rank TrainPipeline loop (collect -> all-gather -> rank 0 update ->
weight broadcast) on CPU ranks (gloo) with a stand-in evaluator and a stand-in trainer -- plumbing only, no HIP."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from alphapig_amd import dist  # noqa: E402
from alphapig_amd.pipeline import TrainPipeline  # noqa: E402
from fakenet import fake_policy_value_batch  # noqa: E402


class TiltNet(object):
    """fakenet's closed-form net, its move probabilities tilted by a learnable vector w (and the value by b): an evaluator
    whose outputs depend on its parameters, so that a broadcast that did not arrive shows in the games played."""

    def __init__(self, hw):
        self._p = {"w": np.zeros(hw, np.float32), "b": np.zeros(1, np.float32)}
        self.sets = 0

    def __call__(self, planes):
        p, v = fake_policy_value_batch(planes)
        p = p.astype(np.float64) * np.exp(self._p["w"].astype(np.float64))[None, :]
        p /= p.sum(axis=1, keepdims=True)
        return p.astype(np.float32), np.tanh(np.arctanh(np.clip(v, -0.999, 0.999)) + self._p["b"][0]).astype(np.float32)

    def policy_value(self, states):
        p, v = self(np.asarray(states))
        return p, v.reshape(-1, 1)

    def params(self):
        return {k: v.copy() for k, v in self._p.items()}

    def set_params(self, prm, _keep_trainer=False):
        self._p = {k: np.asarray(prm[k], np.float32).copy() for k in ("w", "b")}
        self.sets += 1

    def save_model(self, path):
        np.savez(path, **self._p)

    def close(self):
        pass


class TiltTrainer(object):
    """Deterministic stand-in for HipTrainer: moves w towards the batch's mean pi, b towards its mean z."""

    def __init__(self, net):
        self._p = net.params()
        self.t = 0

    def train_step(self, states, pis, zs, lr):
        w = self._p["w"].astype(np.float64)
        sm = np.exp(w - w.max())
        sm /= sm.sum()
        target = np.asarray(pis, np.float64).mean(axis=0)
        self._p["w"] = (w + 50.0 * lr * (target - sm)).astype(np.float32)
        self._p["b"] = (self._p["b"] + np.float32(10.0 * lr * float(np.mean(zs)))).astype(np.float32)
        self.t += 1
        return float(-np.sum(target * np.log(sm + 1e-12))), float(-np.sum(sm * np.log(sm + 1e-12)))

    def get_params(self):
        return {k: v.copy() for k, v in self._p.items()}


def main_real(out_dir):
    """The same loop with the REAL evaluator and the REAL HIP trainer, every rank on cuda:0, collectives on gloo (a one-GPU box
    cannot give each rank its own GPU, and RCCL refuses two ranks on one device): after the run every rank's evaluator
    must answer like rank 0's -- the weights travelled through broadcast_params + set_params."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    rank, world, _ = dist.init(backend="gloo")
    conf = {"board_width": 15, "board_height": 15, "n_in_row": 5, "learn_rate": 1e-3, "lr_multiplier": 1.0, "temp": 1.0,
            "n_playout": 6, "c_puct": 5, "buffer_size": 100000, "batch_size": 32, "epochs": 2, "kl_targ": 0.02,
            "check_freq": 1000, "game_batch_num": 3, "play_batch_size": 2, "pure_mcts_playout_num": 10, "async_update": False,
            "concurrent_games": 16, "n_blocks": 1, "n_filter": 128, "model_dir": os.path.join(out_dir, "models%d" % rank)}
    pipe = TrainPipeline(conf, device=0, seed=5)
    assert pipe.distributed and (pipe.rank, pipe.world) == (rank, world)
    hist = pipe.run()
    planes = (np.random.RandomState(9).rand(6, 9, 15, 15) < 0.2).astype(np.float32)
    p, v = pipe.policy_value_net.policy_value(planes)
    np.savez(os.path.join(out_dir, "real%d.npz" % rank), p=p, v=v)
    res = {"rank": rank, "world": world, "updates": sum(1 for h in hist if "loss" in h),
           "weight_broadcasts": getattr(pipe, "weight_broadcasts", 0), "buffer": len(pipe.data_buffer),
           "has_trainer": getattr(pipe.policy_value_net, "_trainer", None) is not None}
    with open(os.path.join(out_dir, "pipe%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    pipe.close()
    dist.shutdown()


class SlowTiltTrainer(TiltTrainer):
    """... whose optimiser step takes a while (stands in for the GPU time of a real policy_update): what the other ranks
    do meanwhile is what tests/test_pipeline_dist.py looks at."""

    def train_step(self, states, pis, zs, lr):
        import time
        time.sleep(0.6)
        return TiltTrainer.train_step(self, states, pis, zs, lr)


def main_async(out_dir):
    """The asynchronous schedule on CPU ranks: rounds of a few engine steps, a trainer thread on rank 0 whose updates take
    1.2 s each (two epochs of 0.6 s)."""
    rank, world, _ = dist.init(backend="gloo")
    conf = {"board_width": 8, "board_height": 8, "n_in_row": 4, "learn_rate": 2e-3, "lr_multiplier": 1.0, "temp": 1.0,
            "n_playout": 8, "c_puct": 5, "buffer_size": 100000, "batch_size": 16, "epochs": 2, "kl_targ": 1e9,
            "check_freq": 20, "eval_games": 2, "game_batch_num": 60, "play_batch_size": 1, "pure_mcts_playout_num": 10,
            "async_update": True, "round_seconds": 0.15, "concurrent_games": 4,
            "model_dir": os.path.join(out_dir, "models%d" % rank)}
    net = TiltNet(64)
    trainer = SlowTiltTrainer(net) if rank == 0 else None
    pipe = TrainPipeline(conf, policy_value_net=net, seed=77, trainer=trainer, eval_net=TiltNet(64) if rank == 0 else None)
    hist = pipe.run()
    res = {"rank": rank, "world": world, "history": hist, "buffer": len(pipe.data_buffer),
           "weight_broadcasts": pipe.weight_broadcasts, "weights_version": pipe.weights_version,
           "updates_done": pipe.updates_done, "updates_skipped": pipe.updates_skipped,
           "update_intervals": pipe.update_intervals, "round_log": pipe.round_log, "sets": net.sets,
           "trainer_history": getattr(pipe, "trainer_history", []), "taken": pipe._taken,
           "w": net.params()["w"].tolist(), "trainer_w": trainer.get_params()["w"].tolist() if trainer else None,
           "game_indices": sorted(set(int(s.index) for s in pipe.engine.slots if s.active))}
    with open(os.path.join(out_dir, "pipe%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    pipe.engine.close()
    dist.shutdown()


def main():
    out_dir = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == "real":
        return main_real(out_dir)
    if len(sys.argv) > 2 and sys.argv[2] == "async":
        return main_async(out_dir)
    ASYNC = False                 # the lock-step schedule: every rank waits for rank 0's update
    rank, world, _ = dist.init(backend="gloo")
    conf = {"board_width": 8, "board_height": 8, "n_in_row": 4, "learn_rate": 2e-3, "lr_multiplier": 1.0, "temp": 1.0,
            "n_playout": 8, "c_puct": 5, "buffer_size": 100000, "batch_size": 16, "epochs": 2, "kl_targ": 0.02,
            "check_freq": 1000, "game_batch_num": 4, "play_batch_size": 2, "pure_mcts_playout_num": 10, "async_update": ASYNC,
            "concurrent_games": 4, "model_dir": os.path.join(out_dir, "models%d" % rank)}
    net = TiltNet(64)
    trainer = TiltTrainer(net) if rank == 0 else None
    pipe = TrainPipeline(conf, policy_value_net=net, seed=77, trainer=trainer)
    assert pipe.distributed == (world > 1) and (pipe.rank, pipe.world) == (rank, world)
    hist = pipe.run()
    res = {"rank": rank, "world": world, "history": hist, "buffer": len(pipe.data_buffer),
           "weight_broadcasts": getattr(pipe, "weight_broadcasts", 0), "last_gathered": getattr(pipe, "last_gathered", None),
           "lr_multiplier": pipe.lr_multiplier, "sets": net.sets, "train_steps": pipe._train_steps,
           "w": net.params()["w"].tolist(), "b": net.params()["b"].tolist(),
           "trainer_w": trainer.get_params()["w"].tolist() if trainer else None,
           "game_indices": sorted(set(int(s.index) for s in pipe.engine.slots if s.active))}
    with open(os.path.join(out_dir, "pipe%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    pipe.engine.close()
    dist.shutdown()


if __name__ == "__main__":
    main()
