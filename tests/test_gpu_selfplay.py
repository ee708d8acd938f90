"""GPU end-to-end: the batched engine driving the HIP evaluator (BASELINE configs 2 and 3 in
miniature).  Move parity is checked the way SURVEY.md section 7 prescribes: the evaluator outputs
the GPU produced are recorded and replayed into the sequential CPU oracle, so both trees see the
SAME numbers (batched-vs-batch-1 float drift cannot flip a PUCT near-tie), while the numbers
themselves are checked against the float64 net oracle at 1e-4 / 2e-5 elsewhere."""
import os
import random

import numpy as np
import pytest

from alphapig_amd import weights
from alphapig_amd.selfplay import SelfPlayEngine
from oracle import net_ref, selfplay_ref
from oracle.board_ref import RefBoard
from oracle.mcts_ref import RefMCTSPlayer

pytestmark = pytest.mark.gpu


class Recorder(object):
    def __init__(self, net):
        self.net, self.table = net, {}

    def evaluate_codes(self, codes):
        p, v = self.net.evaluate_codes(codes)
        for c, pi, vi in zip(codes, p, v):
            self.table[c.tobytes()] = (pi.copy(), np.float32(vi))
        return p, v


def codes_of(board, stride):
    hw = board.width * board.height
    c = np.zeros(stride, dtype=np.uint8)
    n = len(board.move_list)
    for k, (m, who) in enumerate(zip(board.move_list, board.mover_list)):
        c[m] = (1 if who == board.current_player else 5) + min(n - 1 - k, 3)
    c[hw] = 1 if n % 2 == 0 else 0
    return c.tobytes()


def replay_fn(table, stride):
    def fn(board):
        key = codes_of(board, stride)
        if key not in table:
            # the reference evaluates terminal leaves too and discards the result
            # (mcts_alphaZero.py:124-136); the engine never sends them to the GPU
            assert board.game_end()[0], "oracle visited a non-terminal position the engine never evaluated"
            return zip(board.availables, np.zeros(len(board.availables), np.float32)), np.zeros(1, np.float32)
        p, v = table[key]
        legal = board.availables
        return zip(legal, p[legal]), np.array([v], dtype=np.float32)
    return fn


class Never(object):
    def random(self):
        return 1.0


@pytest.mark.parametrize("cfg", ["simple8", "resnet15", "resnet15-bf16x3", "resnet15-f16x2"])
def test_engine_on_gpu_equals_sequential_oracle_on_recorded_outputs(cfg):
    from alphapig_amd.policy_value_net import PolicyValueNet
    if cfg == "simple8":      # BASELINE config 2 shape: 8x8, 4-in-row, simple net
        w, nrow, npl, G, total = 8, 4, 60, 16, 20
        prm = weights.init_params("simple", 8, 8, 9, seed=1, style="bench")
        net = PolicyValueNet(8, 8, batch_size=64, model_params=prm, net_kind="simple")
        forced = False
    elif cfg == "resnet15":   # BASELINE config 3 shape: 15x15, 5-in-row, residual net
        w, nrow, npl, G, total = 15, 5, 24, 12, 12
        prm = weights.init_params("resnet", 15, 15, 9, 2, 128, seed=2, style="bench")
        net = PolicyValueNet(15, 15, batch_size=64, n_blocks=2, n_filter=128, model_params=prm)
        forced = True
    else:                     # ... on the split trunk kernels: 80 concurrent games = 40-board batches (they take batches > 32)
        w, nrow, npl, G, total = 15, 5, 24, 80, 80
        prm = weights.init_params("resnet", 15, 15, 9, 2, 128, seed=2, style="bench")
        net = PolicyValueNet(15, 15, batch_size=64, n_blocks=2, n_filter=128, model_params=prm, trunk_arith=cfg.split("-")[1])
        forced = True
    rec = Recorder(net)
    eng = SelfPlayEngine(rec, w, w, nrow, n_games=G, n_playout=npl, temp=1.0, base_seed=31337, n_threads=4,
                         pipeline=2, forced_opening=forced)
    eps = eng.play_games(total)
    assert len(eps) >= total and len(rec.table) > 0
    stride = eng.pool.code_stride
    fn = replay_fn(rec.table, stride)
    for e in eps[:6]:
        b = RefBoard(w, w, nrow)
        pl = RefMCTSPlayer(fn, c_puct=5, n_playout=npl, is_selfplay=1, rng=np.random.RandomState(31337 + e.index))
        pyr = random.Random(31337 + e.index) if forced else Never()
        winner, data = selfplay_ref.start_self_play(b, pl, temp=1.0, pyrandom=pyr)
        assert winner == e.winner
        np.testing.assert_array_equal(np.array(b.move_list), e.moves)
        np.testing.assert_allclose(np.stack([d[1] for d in data]), e.pis, rtol=0, atol=1e-12)
        np.testing.assert_array_equal(np.array([d[2] for d in data]), e.zs)
    # and the recorded numbers are the network's: spot-check against the float64 oracle
    keys = list(rec.table.keys())[:32]
    codes = np.stack([np.frombuffer(k, dtype=np.uint8) for k in keys])
    planes = eng.pool.codes_to_planes(codes, 9)
    kind = "simple" if cfg == "simple8" else "resnet"
    o = net_ref.forward(prm, planes, kind, 2, np.float64)
    np.testing.assert_allclose(np.stack([rec.table[k][0] for k in keys]), o[1], rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.array([rec.table[k][1] for k in keys]), o[3][:, 0], rtol=0, atol=2e-5)
    eng.close()
    net.close()


@pytest.mark.parametrize("arith", ["f32", "f16x2"])
def test_full_configuration_engine_equals_sequential_oracle(arith):
    """BASELINE configs[2] EXACTLY -- 15x15, five in a row, n_playout = 400, c_puct 5, temp 1.0, Dirichlet 0.3 / 0.25, the
    10-block / 128-filter residual net (train_mxnet.py:79-91), the forced-opening branch live (game_ai.py:79-111) -- on the
    HIP evaluator, on both trunk arithmetics, against the sequential oracle (mcts_alphaZero.py:141-157, :187-218 and
    game_ai.py:70-139 restated in oracle/mcts_ref.py / selfplay_ref.py, pinned to the reference's own traces on CPU).
    96 concurrent games = 48-board batches, i.e. the BATCHED trunk kernels (trunk15_wino3_kernel / trunk15_wino3h_kernel;
    a board's bits do not depend on the launch shape: tests/test_gpu_net.py).  Game 0 opens freely, game 1 takes the
    forced opening (base seed chosen for that).  Every evaluation those two games consumed is recorded (engine tap) and
    replayed into the oracle: moves, winner and z exact, pi to 1e-12; a 64-row sample of the recorded outputs against the
    float64 net oracle at 2e-5."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    base, G, npl = 1018, 96, 400
    assert [random.Random(base + k).random() < 0.09 for k in (0, 1)] == [False, True]
    prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
    net = PolicyValueNet(15, 15, batch_size=64, n_blocks=10, n_filter=128, model_params=prm, trunk_arith=arith)
    eng = SelfPlayEngine(net, 15, 15, 5, n_games=G, n_playout=npl, c_puct=5, temp=1.0, base_seed=base, n_threads=4,
                         pipeline=2, forced_opening=True)
    chosen, table, batch_sizes = (0, 1), {}, set()

    def tap(ids, codes, p, v):
        batch_sizes.add(len(ids))
        for i, s in enumerate(ids):
            if eng.slots[int(s)].index in chosen:
                table[codes[i].tobytes()] = (p[i].copy(), np.float32(v[i]))
    eng.tap = tap
    steps = 0
    while not set(chosen) <= set(e.index for e in eng.finished):
        eng.run_steps(400)
        steps += 400
        assert steps <= 226 * 400, "a 15x15 game ends within 225 plies"
    assert batch_sizes == {G // 2}                       # every forward was a 48-board batch: the batched kernels
    eps = {e.index: e for e in eng.finished}
    assert eng.stats["forced_openings"] >= 1
    fn = replay_fn(table, eng.pool.code_stride)
    for k in chosen:
        e = eps[k]
        b = RefBoard(15, 15, 5)
        pl = RefMCTSPlayer(fn, c_puct=5, n_playout=npl, is_selfplay=1, rng=np.random.RandomState(base + k))
        winner, data = selfplay_ref.start_self_play(b, pl, temp=1.0, pyrandom=random.Random(base + k))
        assert winner == e.winner
        np.testing.assert_array_equal(np.array(b.move_list), e.moves)
        np.testing.assert_allclose(np.stack([d[1] for d in data]), e.pis, rtol=0, atol=1e-12)
        np.testing.assert_array_equal(np.array([d[2] for d in data]), e.zs)
        if k == 1:                                       # the forced opening: two recorded plies with the one-hot pi
            assert e.pis[0].max() == 0.99999 and e.pis[1].max() == 0.99999
    keys = [list(table.keys())[i] for i in np.random.RandomState(5).permutation(len(table))[:64]]
    codes = np.stack([np.frombuffer(k, dtype=np.uint8) for k in keys])
    o = net_ref.forward(prm, eng.pool.codes_to_planes(codes, 9), "resnet", 10, np.float64)
    np.testing.assert_allclose(np.stack([table[k][0] for k in keys]), o[1], rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.array([table[k][1] for k in keys]), o[3][:, 0], rtol=0, atol=2e-5)
    print("full configuration [%s]: games 0 / 1 = %d / %d plies, %d recorded evaluations, %d engine steps"
          % (arith, len(eps[0].moves), len(eps[1].moves), len(table), steps))
    eng.close()
    net.close()


@pytest.mark.parametrize("arith", ["f32", "f16x2"])
def test_full_size_properties_1024_games(arith):
    """BASELINE config 3 at full width (1024 concurrent games, 10 blocks, n_playout=400) for a few
    rounds, on both trunk arithmetics: size-independent properties -- one leaf per active game per round,
    priors are a distribution, visit counts add up, Q in [-1, 1]."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
    net = PolicyValueNet(15, 15, batch_size=1024, n_blocks=10, n_filter=128, model_params=prm, trunk_arith=arith)
    eng = SelfPlayEngine(net, 15, 15, 5, n_games=1024, n_playout=400, temp=1.0, base_seed=1, pipeline=2)
    rounds = 12
    n = eng.run_steps(rounds)
    assert n == 1024 * rounds
    for g in (0, 511, 1023):
        root = eng.pool.node_children(g, 0)
        assert root["n"] == rounds and int(root["visits"].sum()) == rounds - 1
        assert len(root["acts"]) < 225 or abs(float(root["prior"].sum()) - 1.0) < 1e-4
        assert np.all(np.abs(root["q"]) <= 1.0 + 1e-6)
    eng.close()
    net.close()


def test_full_size_config2_64_games_8x8_simple_net():
    """BASELINE configs[1] at its full size: 64 concurrent 8x8 games (4 in a row), n_playout = 200, the simple
    6-convolution net (policy_value_net_mxnet_simple.py:68-92), for 450 rounds -- every game searches 200 playouts,
    moves and searches on, twice -- with the size-independent properties of the 1024-game test: one leaf per active
    game per round, priors a distribution over the legal moves, visit counts that add up, Q in [-1, 1]; and the
    evaluations the engine consumed are the network's (a sample of recorded rows against the float64 oracle)."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, 9, seed=1, style="bench")
    net = PolicyValueNet(8, 8, batch_size=64, model_params=prm, net_kind="simple")
    rec = Recorder(net)
    eng = SelfPlayEngine(rec, 8, 8, 4, n_games=64, n_playout=200, temp=1.0, base_seed=77, pipeline=2, forced_opening=False)
    n = eng.run_steps(150)
    assert n == 64 * 150                                   # nobody has moved yet: one leaf per game and round
    for g in (0, 31, 63):
        root = eng.pool.node_children(g, 0)
        assert root["n"] == 150 and int(root["visits"].sum()) == 149
        assert len(root["acts"]) == 64 and abs(float(root["prior"].sum()) - 1.0) < 1e-4
        assert np.all(np.abs(root["q"]) <= 1.0 + 1e-6)
    n += eng.run_steps(300)                                # past two moves of every game (200 playouts each, tree re-used)
    assert eng.stats["moves"] >= 2 * 64
    assert 64 * 400 <= n <= 64 * 450                       # at most one evaluation per game and round (terminal leaves need none)
    for g in (0, 63):
        root = eng.pool.node_children(g, 0)
        assert np.all(np.abs(root["q"]) <= 1.0 + 1e-6) and int(root["visits"].sum()) == root["n"] - 1
    keys = list(rec.table.keys())
    pick = [keys[i] for i in np.random.RandomState(3).permutation(len(keys))[:48]]
    codes = np.stack([np.frombuffer(k, dtype=np.uint8) for k in pick])
    o = net_ref.forward(prm, eng.pool.codes_to_planes(codes, 9), "simple", dtype=np.float64)
    np.testing.assert_allclose(np.stack([rec.table[k][0] for k in pick]), o[1], rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.array([rec.table[k][1] for k in pick]), o[3][:, 0], rtol=0, atol=2e-5)
    eng.close()
    net.close()


def test_closed_loop_selfplay_train_selfplay():
    """One AlphaZero iteration on the GPU: HIP self-play -> augmentation -> (interim torch)
    train_step -> re-folded weights in the HIP evaluator -> the evaluator now agrees with the
    oracle forward on the UPDATED parameters."""
    pytest.importorskip("torch")
    from alphapig_amd.augment import get_equi_data
    from alphapig_amd.policy_value_net import PolicyValueNet
    from alphapig_amd.selfplay import episodes_to_tuples
    prm = weights.init_params("resnet", 8, 8, 9, 2, 64, seed=4, style="reference")
    net = PolicyValueNet(8, 8, batch_size=64, n_blocks=2, n_filter=64, model_params=prm)
    eng = SelfPlayEngine(net, 8, 8, 4, n_games=16, n_playout=20, temp=1.0, base_seed=11, n_threads=4, pipeline=2,
                         forced_opening=False)
    eps = eng.play_games(16)
    states, pis, zs = episodes_to_tuples(eps, eng.pool)
    data = get_equi_data(list(zip(states, pis, zs)), 8, 8)
    rs = np.random.RandomState(0)
    pick = rs.permutation(len(data))[:64]
    batch = [data[i] for i in pick]
    sb = np.stack([b[0] for b in batch]); pb = np.stack([b[1] for b in batch]); zb = np.array([b[2] for b in batch])
    before = net.policy_value(sb)[0]
    losses = [float(net.train_step(sb, pb, zb, 2e-3)[0][0]) for _ in range(5)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    after = net.policy_value(sb)[0]
    assert np.abs(after - before).max() > 1e-6                      # the evaluator really got new weights
    new_prm = net.params()
    arg, aux = net.get_policy_param()                              # the reference's pair (policy_value_net_mxnet.py:301)
    assert set(arg) | set(aux) == set(new_prm) and all(k.endswith(("_mean", "_var")) for k in aux)
    o = net_ref.forward(new_prm, sb, "resnet", 2, np.float64)
    np.testing.assert_allclose(after, o[1], rtol=0, atol=2e-5)
    eps2 = eng.play_games(eng.stats["games"] + 4)                   # and self-play goes on with them
    assert len(eps2) >= 20
    eng.close()
    net.close()


def test_train_pipeline_runs_a_few_batches():
    """TrainPipeline (reference train_mxnet.py:38-300 wired onto the engine): a few batches of
    self-play + update + one arena evaluation complete and produce finite numbers."""
    pytest.importorskip("torch")
    from alphapig_amd.pipeline import TrainPipeline
    conf = dict(board_width=8, board_height=8, n_in_row=4, learn_rate=2e-3, lr_multiplier=1.0, temp=1.0,
                n_playout=30, c_puct=5, buffer_size=5000, batch_size=64, epochs=2, kl_targ=0.02, check_freq=3,
                pure_mcts_playout_num=30, game_batch_num=3, play_batch_size=2, concurrent_games=16, n_blocks=1,
                n_filter=64, eval_games=4, model_dir="/tmp/apz_models_test", async_update=False)
    tp = TrainPipeline(conf, seed=3)
    hist = tp.run()
    assert len(hist) == 3 and "win_ratio" in hist[-1] and 0.0 <= hist[-1]["win_ratio"] <= 1.0
    assert any("loss" in h and np.isfinite(h["loss"]) for h in hist)
    assert tp._taken == 6 and len(tp.data_buffer) > 0
    tp.close()


def test_asynchronous_train_pipeline_on_the_gpu():
    """The asynchronous schedule with the REAL pieces on one GPU (alphapig_amd/pipeline.py: train_mxnet.py:265-300 with
    the update in a trainer thread): self-play rounds keep stepping while the HIP trainer updates on its own stream with
    its own evaluator handle; every published version is installed into the self-play evaluator device to device, the
    arena and a checkpoint run in the trainer thread, and after the final drain the self-play evaluator answers exactly
    like the trainer's own."""
    pytest.importorskip("torch")
    from alphapig_amd.pipeline import TrainPipeline
    conf = dict(board_width=15, board_height=15, n_in_row=5, learn_rate=1e-3, lr_multiplier=1.0, temp=1.0,
                n_playout=12, c_puct=5, buffer_size=200000, batch_size=64, epochs=3, kl_targ=0.02, check_freq=25,
                pure_mcts_playout_num=20, game_batch_num=60, play_batch_size=1, concurrent_games=64, n_blocks=2,
                n_filter=128, eval_games=2, model_dir="/tmp/apz_models_async", async_update=True, round_seconds=0.1)
    tp = TrainPipeline(conf, seed=5)
    steps_log, run_steps = [], tp.engine.run_steps

    def logged_run_steps(*a, **k):
        import time
        t0 = time.time()
        r = run_steps(*a, **k)
        steps_log.append((t0, time.time()))
        return r
    tp.engine.run_steps = logged_run_steps
    hist = tp.run()
    assert hist[-1]["games_collected"] >= 60 and tp._taken == hist[-1]["games_collected"]
    assert tp.updates_done >= 2 and tp.updates_done + tp.updates_skipped <= hist[-1]["games_collected"]
    versions = [h["version"] for h in hist]
    assert versions == sorted(versions) and versions[-1] == tp.updates_done == tp.weights_version
    assert tp.weight_broadcasts == len(set(v for v in versions if v > 0))
    # self-play went on during the updates: engine steps (8 scheduler rounds per call) ran INSIDE update intervals
    inside = sum(1 for a, b in tp.update_intervals for t0, t1 in steps_log if a < t0 and t1 < b)
    assert inside >= 1, (tp.update_intervals[:3], steps_log[:5])
    ups = [h for h in tp.trainer_history if "loss" in h]
    assert ups and all(np.isfinite(h["loss"]) and np.isfinite(h["kl"]) for h in ups)
    assert any("win_ratio" in h for h in tp.trainer_history) and os.path.exists("/tmp/apz_models_async/current_policy.model")
    planes = (np.random.RandomState(2).rand(5, 9, 15, 15) < 0.2).astype(np.float32)
    pa, va = tp.policy_value_net.policy_value(planes)          # installed from the last published snapshot
    tp._async_trainer.sync_evaluator(tp._eval_net)
    pb, vb = tp._eval_net.policy_value(planes)                 # the trainer's own evaluator on the trainer's weights
    assert np.array_equal(pa, pb) and np.array_equal(va, vb)
    tp.close()


def test_rccl_collectives_world_size_1():
    """backend nccl (= RCCL) with one rank on the GPU: count gather + padded tuple all-gather."""
    pytest.importorskip("torch")
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(here, "_rccl_worker.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "rccl world_size=1 ok" in r.stdout, r.stdout[-2000:]


def test_two_rank_bench_rehearsal_on_one_gpu():
    """`bench.py --gpus 2` through its own rank spawner with the REAL evaluator: both ranks on cuda:0, collectives on gloo
    (RCCL refuses two ranks on one device; RCCL itself runs at world size 1 in the tests around this one).  What a one-GPU
    box can show of configs[3]: sharded games, the tuple all-gather with real payloads, per-rank rates, one JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5", "--games", "256",
           "--rehearse-on-one-gpu", "--no-extras", "--prewarm-s", "0.2", "--deadline-s", "600"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["launcher"] == "self-spawned" and d["valid"] is False
    assert len(d["per_rank_leaf_evals_per_s"]) == 2 and min(d["per_rank_leaf_evals_per_s"]) > 1000
    assert d["leaf_evals_per_s"] > 2000 and d["config"]["games_per_gpu"] == 256 and d["roofline"]["frac"] > 0.0   # (the two ranks' kernels share the GPU)
    # the N > 1 line measures the round's exchange with a payload: 256 games x mean plies rows per rank, both ranks' rows verified
    ex = d["exchange"]
    assert ex["payload_verified"] is True and ex["ranks_seen"] == 2 and ex["rows_per_rank"] == int(round(256 * d["config"]["mean_plies_per_game"]))
    assert ex["bytes_gathered_per_rank"] == 2 * ex["rows_per_rank"] * 1144 and ex["ms"] > 0 and "cpu_baseline" in d


def test_two_rank_training_pipeline_with_the_real_trainer(tmp_path):
    """configs[4]'s loop on TWO ranks with the real evaluator and the real HIP trainer (both ranks on cuda:0, collectives on
    gloo -- see test_two_rank_bench_rehearsal_on_one_gpu): rank 0 trains, rank 1 never builds a trainer, and after the run
    both evaluators give the same answers: the weights arrived through dist.broadcast_params."""
    import json
    import os
    import random
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(28500 + random.randint(0, 1500)), os.path.join(here, "_pipeline_worker.py"), str(tmp_path), "real"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    r0, r1 = (json.load(open(tmp_path / ("pipe%d.json" % k))) for k in (0, 1))
    assert r0["updates"] >= 1 and r0["weight_broadcasts"] == r1["weight_broadcasts"] == r0["updates"]
    assert r0["has_trainer"] and not r1["has_trainer"] and r1["buffer"] == 0 and r0["buffer"] > 0
    a, b = np.load(tmp_path / "real0.npz"), np.load(tmp_path / "real1.npz")
    np.testing.assert_allclose(a["p"], b["p"], rtol=0, atol=1e-6)     # rank 0: refreshed device to device; rank 1: folded on the host
    np.testing.assert_allclose(a["v"], b["v"], rtol=0, atol=1e-6)


def test_rccl_training_pipeline_world_size_1():
    """configs[4]'s loop with the real HIP trainer and evaluator, collectives on RCCL (one rank): self-play ->
    all_gather_tuples -> policy_update (train_mxnet.py:194-240) -> flat weight broadcast -> load_device_params
    (policy_value_net_mxnet.py:295-297 across ranks).  The 2-rank plumbing runs on gloo in tests/test_pipeline_dist.py."""
    pytest.importorskip("torch")
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_PORT"] = "29613"
    r = subprocess.run([sys.executable, os.path.join(here, "_rccl_worker.py"), "pipeline"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "rccl pipeline ok" in r.stdout, r.stdout[-3000:]


def test_arena_on_gpu_equals_sequential_oracle_on_recorded_outputs():
    """SURVEY 8f rank 3 on the GPU: `policy_evaluate` (train_mxnet.py:242-263 -> Game.start_play, game.py:204-230)
    as M concurrent matches through the HIP evaluator.  The evaluator outputs the GPU produced are recorded and
    replayed into the sequential oracle players (RefMCTSPlayer in play mode against RefPureMCTSPlayer, one legacy
    MT19937 stream per match as after np.random.seed(base + i)): every match must come out move for move."""
    from alphapig_amd.arena import Arena, win_ratio
    from alphapig_amd.policy_value_net import PolicyValueNet
    from oracle.mcts_ref import RefPureMCTSPlayer
    w, nrow, npl, pure_n, base, n = 15, 5, 40, 30, 4242, 6
    prm = weights.init_params("resnet", 15, 15, 9, 2, 128, seed=6, style="bench")
    net = PolicyValueNet(15, 15, batch_size=64, n_blocks=2, n_filter=128, model_params=prm)
    rec = Recorder(net)
    res = Arena(rec, w, w, nrow, n_playout=npl, pure_mcts_playout_num=pure_n, base_seed=base, n_threads=4,
                max_concurrent=4).play(n)                                   # two blocks: 4 + 2 matches
    assert [r.index for r in res] == list(range(n)) and len(rec.table) > 0
    stride = 240
    fn = replay_fn(rec.table, stride)
    for r in res:
        b = RefBoard(w, w, nrow)
        b.init_board(r.index % 2)
        rs = np.random.RandomState(base + r.index)
        players = {1: RefMCTSPlayer(fn, 5, npl, 0, rng=rs), 2: RefPureMCTSPlayer(5, pure_n, rng=rs)}
        while True:
            b.do_move(int(players[b.get_current_player()].get_action(b)))
            end, winner = b.game_end()
            if end:
                break
        assert list(r.moves) == list(b.move_list), r.index
        assert r.winner == winner
    assert 0.0 <= win_ratio(res) <= 1.0
    net.close()


def test_config5_slice_1024_games_n_playout_1600():
    """BASELINE config 5's per-GPU slice (mcts_alphaZero.py:147-149 with n_playout = 1600; 1024 concurrent games,
    10-block net, temperature schedule): a few hundred scheduler rounds at full size.  Size-independent properties:
    every round evaluates one leaf per active game, playouts are conserved (net evaluations + terminal playouts =
    rounds), no game moves before its 1600 playouts are done, the move that is then played is a legal root child,
    the tree arenas were pre-touched (31 GB: the memory-aware limit, not the old 16 GB cliff) and no tree outgrew
    its reservation."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    G, npl, rounds = 1024, 1600, 1640
    prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
    net = PolicyValueNet(15, 15, batch_size=G // 2, n_blocks=10, n_filter=128, model_params=prm)
    eng = SelfPlayEngine(net, 15, 15, 5, n_games=G, n_playout=npl, c_puct=5, temp=1.0, base_seed=555, pipeline=2,
                         temp_schedule=[(0, 1.0), (30, 0.1)])
    info = eng.pool.pool_info()
    assert info["arena_bytes"] > 25e9 and info["pretouched"]
    leafs = eng.run_steps(1500)
    assert leafs == G * 1500                                        # one evaluated leaf per game and round
    assert eng.stats["moves"] <= G // 100                           # (a move before round 1600 needs terminal-leaf playouts)
    leafs += eng.run_steps(rounds - 1500)                          # crosses the first searched move of every game
    assert leafs == G * rounds and eng.stats["moves"] >= G
    for s in range(0, G, 97):
        mv, _ = eng.pool.history(s)
        assert len(mv) >= 1 and len(set(mv.tolist())) == len(mv)    # legal: no cell played twice
        st = eng.pool.stats(s)
        assert st["net_evals"] == rounds and st["net_evals"] + st["terminal_playouts"] >= npl
    cap = (npl + 8) * 226 * 5 // 4
    assert eng.pool.pool_info()["peak_nodes"] <= cap
    eng.close()
    net.close()


def test_laned_evaluator_overlapping_streams_play_the_same_games():
    """policy_value_net.LanedEvaluator: one engine handle (own HIP stream, own activation buffers) per pipeline group, so
    the groups' forwards overlap on the GPU (BASELINE configs[1]: 8x8, simple net, policy_value_net_mxnet_simple.py:68-92).
    A position's outputs do not depend on the lane, so the games are the single-lane engine's games, bit for bit."""
    from alphapig_amd.policy_value_net import LanedEvaluator, PolicyValueNet
    prm = weights.init_params("simple", 8, 8, 9, seed=1, style="bench")
    out = []
    for lanes, pipeline in ((1, 2), (2, 2), (2, 4)):
        net = PolicyValueNet(8, 8, batch_size=32, model_params=prm, net_kind="simple")
        ev = LanedEvaluator.like(net, lanes) if lanes > 1 else net
        eng = SelfPlayEngine(ev, 8, 8, 4, n_games=16, n_playout=40, temp=1.0, base_seed=4242, n_threads=2, pipeline=pipeline,
                             forced_opening=False)
        eps = eng.play_games(12)
        out.append(eps)
        eng.close()
        ev.close()
    for other in out[1:]:
        assert [e.index for e in other] == [e.index for e in out[0]]
        for a, b in zip(out[0], other):
            np.testing.assert_array_equal(a.moves, b.moves)
            np.testing.assert_array_equal(a.pis, b.pis)
            np.testing.assert_array_equal(a.zs, b.zs)
            assert a.winner == b.winner
