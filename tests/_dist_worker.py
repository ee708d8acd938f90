"""Worker for tests/test_dist_gloo.py: run under torch.distributed.run with 2 CPU ranks (gloo)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from alphapig_amd import dist  # noqa: E402
from alphapig_amd.selfplay import SelfPlayEngine  # noqa: E402
from fakenet import fake_policy_value_batch  # noqa: E402


def main():
    out_dir = sys.argv[1]
    total = int(sys.argv[2])
    rank, world, _ = dist.init(backend="gloo")
    mine = dist.shard_indices(total, rank, world)
    eng = SelfPlayEngine(fake_policy_value_batch, 8, 8, 4, n_games=2, n_playout=16, temp=1.0, base_seed=555,
                         n_threads=1, pipeline=2, forced_opening=False, index_offset=rank, index_stride=world)
    eps = eng.play_games(len(mine))
    assert [e.index for e in eps] == mine, (rank, [e.index for e in eps], mine)
    codes = np.concatenate([e.codes for e in eps])
    pis = np.concatenate([e.pis for e in eps]).astype(np.float32)
    zs = np.concatenate([e.zs for e in eps]).astype(np.float32)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), codes=codes, pis=pis, zs=zs,
             idx=np.array([e.index for e in eps]), lens=np.array([len(e.moves) for e in eps]))
    g_codes, g_pis, g_zs = dist.all_gather_tuples(codes, pis, zs)
    np.savez(os.path.join(out_dir, "gathered%d.npz" % rank), codes=g_codes, pis=g_pis, zs=g_zs)
    tmax = dist.all_reduce_max(float(rank + 1))
    tsum = dist.all_reduce_sum(float(rank + 1))
    assert tmax == world and tsum == world * (world + 1) / 2
    # an empty contribution from one rank must work too
    e_codes, e_pis, e_zs = dist.all_gather_tuples(codes[:0] if rank == 0 else codes, pis[:0] if rank == 0 else pis,
                                                  zs[:0] if rank == 0 else zs)
    np.savez(os.path.join(out_dir, "gathered_empty%d.npz" % rank), codes=e_codes, pis=e_pis, zs=e_zs)
    # the round's exchange at FULL size (BASELINE configs[3]: 1024 games x 68.4 plies per rank = 80 MB per rank), timed
    if len(sys.argv) > 3:
        import json
        ex = dist.measure_exchange(int(sys.argv[3]), repeats=1)      # 15x15 rows: 240 + 900 + 4 bytes
        if rank == 0:
            with open(os.path.join(out_dir, "exchange.json"), "w") as f:
                json.dump(ex, f)
    dist.barrier()
    eng.close()


if __name__ == "__main__":
    main()
