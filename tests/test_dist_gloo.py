"""N > 1 path on CPU: 2 gloo ranks shard the games and all-gather the finished tuples."""
import os
import random
import subprocess
import sys

import numpy as np

from alphapig_amd import dist
from alphapig_amd.selfplay import SelfPlayEngine
from fakenet import fake_policy_value_batch


def test_shard_indices():
    assert dist.shard_indices(10, 0, 4) == [0, 4, 8]
    assert dist.shard_indices(10, 3, 4) == [3, 7]
    assert sorted(sum((dist.shard_indices(10, r, 4) for r in range(4)), [])) == list(range(10))


def test_single_process_passthrough():
    c = np.zeros((3, 80), np.uint8)
    p = np.ones((3, 64), np.float32)
    z = np.array([1, -1, 0], np.float32)
    g = dist.all_gather_tuples(c, p, z)
    assert g[0].shape == (3, 80) and g[1].shape == (3, 64) and list(g[2]) == [1, -1, 0]
    assert dist.all_reduce_max(3.5) == 3.5 and dist.all_reduce_sum(2.0) == 2.0


def test_two_rank_gloo_allgather(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    total = 5
    port = 29500 + random.randint(0, 2000)
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(here, "_dist_worker.py"),
           str(tmp_path), str(total)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert list(r0["idx"]) == [0, 2, 4] and list(r1["idx"]) == [1, 3]
    for k in ("codes", "pis", "zs"):
        want = np.concatenate([r0[k], r1[k]])
        for rank in (0, 1):
            got = np.load(tmp_path / ("gathered%d.npz" % rank))[k]
            np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(np.load(tmp_path / "gathered_empty1.npz")[k], r1[k])
    # sharded games are the same games a single engine plays (seed = base + global index)
    eng = SelfPlayEngine(fake_policy_value_batch, 8, 8, 4, n_games=3, n_playout=16, temp=1.0, base_seed=555,
                         n_threads=1, pipeline=1, forced_opening=False)
    eps = eng.play_games(total)
    lens = {int(i): int(l) for r in (r0, r1) for i, l in zip(r["idx"], r["lens"])}
    assert [len(e.moves) for e in eps] == [lens[i] for i in range(total)]
    np.testing.assert_array_equal(np.concatenate([eps[i].codes for i in (0, 2, 4)]), r0["codes"])
    eng.close()


def test_two_rank_exchange_at_full_round_size(tmp_path):
    """The exchange with a payload: one all_gather_tuples of a whole round -- 1024 games x 68.39 plies = 70 031
    rows x 1 144 B = 80 MB per rank -- between two gloo ranks; every rank checks every rank's rows on arrival
    (`payload_verified`) and the figures bench.py prints are there."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    port = 25500 + random.randint(0, 2000)
    rows = int(round(1024 * 68.39))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(here, "_dist_worker.py"),
           str(tmp_path), "2", str(rows)]
    r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    ex = json.load(open(tmp_path / "exchange.json"))
    assert ex["payload_verified"] is True and ex["ranks_seen"] == 2 and ex["backend"] == "gloo"
    assert ex["rows_per_rank"] == rows and ex["row_bytes"] == 240 + 4 * 225 + 4 and ex["bytes_sent_per_rank"] > 80e6
    assert ex["bytes_sent_per_rank"] == rows * ex["row_bytes"] and ex["bytes_gathered_per_rank"] == 2 * rows * ex["row_bytes"]
    assert ex["ms"] >= ex["collective_ms"] > 0 and ex["GB_per_s_collective"] > 0


def test_bench_multi_rank_plumbing(tmp_path):
    """`torchrun ... bench.py --gpus 2` end to end on CPU ranks (gloo + stand-in evaluator):
    argument handling, rank sharding, barriers, MAX/SUM reductions, the tuple all-gather and the
    single JSON line from rank 0."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 31500 + random.randint(0, 2000)
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "6", "--warmup", "2", "--plumbing-test"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["valid"] is False and d["vs_baseline"] is None and d["unit"] == "games/s"
    assert d["leaf_evals_per_s"] > 0 and d["value"] > 0
    assert d["config"]["games_per_gpu"] == 8
    assert d["ranks_seen"] == 2 and d["host_threads_per_rank"] >= 1 and d["launcher"] == "torchrun"
    # the N > 1 line measures the exchange with a payload (a full round's rows per rank) and points at the CPU baseline
    ex = d["exchange"]
    assert ex["payload_verified"] is True and ex["ranks_seen"] == 2 and ex["backend"] == "gloo"
    assert ex["rows_per_rank"] == int(round(8 * d["config"]["mean_plies_per_game"])) and ex["row_bytes"] == 240 + 900 + 4
    assert ex["bytes_gathered_per_rank"] == 2 * ex["rows_per_rank"] * ex["row_bytes"] and ex["collective_ms"] > 0
    assert "cpu_baseline" in d and "N = 1" in d["cpu_baseline"]["note"] and "parity_pin" in d


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher (the shape of the driver's N = 1 command): the parent starts
    the two ranks itself, rank 0's single JSON line comes through, and it says so (n_gpus, ranks_seen from an
    all-reduce, host threads per rank from the CPU share / 2)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--plumbing-test"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["launcher"] == "self-spawned"
    assert d["host_threads_per_rank"] == max(1, min(16, d["host_cpu_share"] // 2))
    assert d["tree_arena_gb_per_rank"] > 0 and d["valid"] is False
    assert len(d["per_rank_leaf_evals_per_s"]) == 2 and all(x > 0 for x in d["per_rank_leaf_evals_per_s"])
    assert isinstance(d["warnings"], list)
    # a rank that dies takes the whole command down with a non-zero exit code
    bad = subprocess.run(cmd, env=dict(env, APZ_BENCH_TEST_FAIL_RANK="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=120)
    assert bad.returncode != 0 and "{" not in bad.stdout


def test_bench_supervisor_deadlines():
    """The self-spawning supervisor must not wait for ever: a rank that hangs before the rendezvous (the others then sit
    in init_process_group), and a rank that leaves with exit code 0 while the others wait for it, both end in a
    non-zero exit with every rank killed, within the configured deadlines."""
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--plumbing-test"]
    t0 = time.time()
    hung = subprocess.run(base + ["--rank-silence-s", "6"], env=dict(env, APZ_BENCH_TEST_HANG_RANK="1"),
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert hung.returncode != 0 and "{" not in hung.stdout and "no rank wrote to its log" in hung.stderr
    assert time.time() - t0 < 120
    t0 = time.time()
    late = subprocess.run(base + ["--deadline-s", "5", "--rank-silence-s", "600"], env=dict(env, APZ_BENCH_TEST_HANG_RANK="1"),
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert late.returncode != 0 and "deadline" in late.stderr and time.time() - t0 < 120
    early = subprocess.run(base + ["--early-exit-grace-s", "4"], env=dict(env, APZ_BENCH_TEST_EARLY_EXIT_RANK="1"),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert early.returncode != 0 and "exited 0 but the others are still running" in early.stderr


def test_eight_rank_rehearsal(tmp_path):
    """BASELINE configs 4 / 5 shard the games over the 8 GPUs of a node.  The same code on 8 CPU ranks (gloo): every
    global game index is played by exactly one rank (round-robin), with the seed of its index, and every rank ends
    up with the same gathered tuples, rank-major."""
    here = os.path.dirname(os.path.abspath(__file__))
    total, world = 19, 8
    port = 25500 + random.randint(0, 2000)
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(here, "_dist_worker.py"),
           str(tmp_path), str(total)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    ranks = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(world)]
    assert sorted(int(i) for rk in ranks for i in rk["idx"]) == list(range(total))
    for k, rk in enumerate(ranks):
        assert list(rk["idx"]) == list(range(k, total, world))
    for key in ("codes", "pis", "zs"):
        want = np.concatenate([rk[key] for rk in ranks])
        for k in range(world):
            np.testing.assert_array_equal(np.load(tmp_path / ("gathered%d.npz" % k))[key], want)
    # the same games as ONE engine plays them
    eng = SelfPlayEngine(fake_policy_value_batch, 8, 8, 4, n_games=4, n_playout=16, temp=1.0, base_seed=555,
                         n_threads=1, pipeline=1, forced_opening=False)
    eps = eng.play_games(total)
    lens = {int(i): int(l) for rk in ranks for i, l in zip(rk["idx"], rk["lens"])}
    assert [len(e.moves) for e in eps] == [lens[i] for i in range(total)]
    eng.close()


def test_bench_eight_self_spawned_ranks():
    """`python bench.py --gpus 8` (the driver's N = 8 command shape) on CPU ranks: eight ranks come up, one line."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--plumbing-test"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["launcher"] == "self-spawned" and d["scaling"] == "weak"
    assert d["host_threads_per_rank"] == max(1, min(16, d["host_cpu_share"] // 8))


def test_bench_exchange_probe_child_process_protocol():
    """The N = 1 line's `exchange` object: bench.exchange_probe_world1 forms a one-rank process group in a CHILD process (gloo
    here, RCCL on the GPU box), runs dist.measure_exchange and hands the verified figures back; a child that does not
    answer in time costs this object, not the line."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    ex = bench.exchange_probe_world1(3000, timeout_s=120.0)
    assert ex.get("payload_verified") is True and ex["ranks_seen"] == 1 and ex["rows_per_rank"] == 3000
    assert ex["bytes_sent_per_rank"] == 3000 * 1144 and ex["process"].startswith("child process")
    late = bench.exchange_probe_world1(3000, timeout_s=0.01)
    assert "error" in late and "did not answer" in late["error"]


def test_block_copy_parallel_slices_and_page_touch():
    """dist._block_copy (the exchange's pack / unpack of a round's 80 MB: large blocks in parallel slices) and
    dist._touch_pages (first touch of the arrays the consumer keeps, on a helper thread under the collective): byte-exact for
    sizes around the slice boundaries, and the touch leaves shapes alone."""
    from alphapig_amd import dist
    rs = np.random.RandomState(3)
    for n in (0, 1, 4095, (8 << 20) - 1, (8 << 20) + 1, (9 << 20) + 4097, 20 << 20):
        src = rs.randint(0, 256, n).astype(np.uint8)
        dst = np.empty(n, np.uint8)
        dist._block_copy(dst, src)
        assert np.array_equal(dst, src), n
    a, b, c = np.empty((1000, 225), np.float32), np.empty((0, 240), np.uint8), np.empty(7, np.float32)
    dist._touch_pages((a, b, c))
    assert a.shape == (1000, 225) and b.shape == (0, 240) and c.shape == (7,)
