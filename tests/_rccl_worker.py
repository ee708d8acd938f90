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
dist.barrier()
td.destroy_process_group()
print("rccl world_size=1 ok")
