"""RCCL path at world_size = 1 on the GPU box (SURVEY.md 8e testability): the same collectives the
8-GPU run uses, on device tensors, through backend "nccl"."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphapig_amd import dist  # noqa: E402

os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
rank, world, local = dist.init(backend="nccl", force=True)
import torch.distributed as td  # noqa: E402
assert td.is_initialized() and td.get_backend() == "nccl" and world == 1
rs = np.random.RandomState(0)
codes = rs.randint(0, 9, size=(37, 240)).astype(np.uint8)
pis = rs.rand(37, 225).astype(np.float32)
zs = rs.choice([-1.0, 0.0, 1.0], size=37).astype(np.float32)
g = dist.all_gather_tuples(codes, pis, zs)
assert np.array_equal(g[0], codes) and np.array_equal(g[1], pis) and np.array_equal(g[2], zs)
e = dist.all_gather_tuples(codes[:0], pis[:0], zs[:0])
assert e[0].shape[0] == 0
assert dist.all_reduce_max(2.5) == 2.5 and dist.all_reduce_sum(4.0) == 4.0
assert dist.all_gather_floats(1.5) == [1.5] and dist.broadcast_floats([1.0, 2.0]) == [1.0, 2.0]
# the exchange at full round size (1024 games x 68.39 plies = 80 MB) through RCCL on device buffers: bench.py's `exchange`
ex = dist.measure_exchange(int(round(1024 * 68.39)), repeats=2)
assert ex["payload_verified"] is True and ex["backend"] == "nccl" and ex["bytes_sent_per_rank"] > 80e6 and ex["collective_ms"] > 0
print("exchange (RCCL, world size 1):", ex["ms"], "ms whole call,", ex["collective_ms"], "ms collective")

if len(sys.argv) > 1 and sys.argv[1] == "pipeline":
    # The multi-rank training loop (alphapig_amd/pipeline.py) with the REAL trainer and evaluator, collectives on RCCL:
    # self-play -> all-gather of [codes | pi | z] -> policy_update on the HIP trainer -> flat weight broadcast.
    import torch
    from alphapig_amd import weights
    from alphapig_amd.pipeline import TrainPipeline
    from alphapig_amd.policy_value_net import PolicyValueNet
    conf = {"board_width": 15, "board_height": 15, "n_in_row": 5, "learn_rate": 1e-3, "lr_multiplier": 1.0, "temp": 1.0,
            "n_playout": 6, "c_puct": 5, "buffer_size": 100000, "batch_size": 32, "epochs": 2, "kl_targ": 0.02,
            "check_freq": 1000, "game_batch_num": 3, "play_batch_size": 2, "pure_mcts_playout_num": 10, "async_update": False,
            "concurrent_games": 16, "n_blocks": 1, "n_filter": 128}
    pipe = TrainPipeline(conf, device=0, seed=3)
    assert pipe.distributed and pipe.world == 1
    hist = pipe.run()
    ups = [h for h in hist if "loss" in h]
    assert ups and pipe.weight_broadcasts == len(ups) and pipe.last_gathered > 0 and all(np.isfinite(h["loss"]) for h in ups)
    # what a rank > 0 does with the broadcast: the flat RCCL buffer, split back into tensors, refreshes an evaluator device
    # to device -- and that evaluator then answers exactly like rank 0's (which was refreshed from the trainer directly)
    tr = pipe.policy_value_net._trainer
    got = dist.broadcast_params({k: tr.p[k] for k in sorted(tr.p)}, src=0)
    assert all(v.is_cuda and v.dtype == torch.float32 for v in got.values())
    for k in tr.p:
        assert torch.equal(got[k], tr.p[k]), k
    other = PolicyValueNet(15, 15, batch_size=32, n_blocks=1, n_filter=128,
                           model_params=weights.init_params("resnet", 15, 15, 9, 1, 128, seed=99, style="bench"))
    other.load_device_params(got)
    planes = (np.random.RandomState(5).rand(8, 9, 15, 15) < 0.2).astype(np.float32)
    pa, va = pipe.policy_value_net.policy_value(planes)
    pb, vb = other.policy_value(planes)
    assert np.array_equal(pa, pb) and np.array_equal(va, vb)
    other.close()
    pipe.close()
    print("rccl pipeline ok: %d updates, %d tuples in the last all-gather" % (len(ups), pipe.last_gathered))

dist.barrier()
td.destroy_process_group()
print("rccl world_size=1 ok")
