#!/usr/bin/env python3
"""bench.py -- self-play throughput of the batched MI355X engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2]; configs[3] = the same per GPU sharded over 8): 15x15 board,
5-in-row, n_playout=400, c_puct=5, temp=1.0, Dirichlet 0.3/0.25, 10-block / 128-filter residual
net (random 'bench' init, synthetic: there are no published weights), 1024 concurrent games per
GPU.  One "step" = one pass of the hot path over one batch: every one of the 1024 games runs
select -> (leaf) -> expand/backup for one playout and the 1024 leaves are evaluated on the GPU
(in `pipeline` coalesced sub-batches so host tree work overlaps the kernels).

Reported `value` = self-play games/s = (playouts completed in the timed region / n_playout)
plies / (mean plies per game) / seconds, aggregated over ranks (weak scaling: 1024 games per
GPU).  Mean plies per game is a property of the workload, measured by playing complete games
with this exact configuration (`--full-games` / `--count-games`, cached in profiles/calibration_r*.json);
`leaf_evals_per_s` (directly counted) is printed alongside.  `--count-games SECONDS` counts
finished games directly in a steady-state window (continuous refill; minutes).

`--gpus N` without a torchrun environment starts the N ranks itself (fresh child processes, one per
GPU, RCCL over 127.0.0.1); under `python -m torch.distributed.run` the ranks are the launcher's.

Trunk arithmetic (round 6): batches of more than 32 boards run the 128 -> 128 trunk convolutions on the fp16 matrix pipe
with every fp32 operand split into two fp16 terms and fp32 accumulation (csrc/trunk15_wino3h.h, `--trunk-arith f16x2`, the
default "auto"): <= 1e-4 on the logits against the float64 oracle like the exact kernel, different low-order bits.
`--trunk-arith f32` runs the exact-fp32 kernel; the default line carries its rate as `value_exact_f32`.

`value` is COUNTED inside this run when the `counted` leg succeeds (default command, N = 1: a child process plays with
continuous refill until every slot has finished a game, then counts finished games over 60 s); `value_derived` keeps the
rate derived from the timed region.

Extra objects on the JSON line: `roofline` (dominant kernel = trunk 3x3 conv; `frac` = EXECUTED MFMA flops / time / peak
of the pipe it runs on), `roofline_stem` (north_star's HBM target shape 8192x4x15x15 and the real C_in=9 stem),
`cpu_baseline` (sequential CPU oracle port, bounded sample).
"""
import argparse
import json
import os
import sys
import time

if "GOMP_SPINCOUNT" not in os.environ and "OMP_WAIT_POLICY" not in os.environ:
    os.environ["GOMP_SPINCOUNT"] = "100000"   # before numpy / torch load an OpenMP runtime (see alphapig_amd/_native.py)

import numpy as np  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from alphapig_amd import dist, weights  # noqa: E402

H = W = 15
N_IN_ROW = 5
N_PLAYOUT = 400
N_BLOCKS = 10
N_FILTER = 128
GAMES_PER_GPU = 1024
CALIB = next((p_ for p_ in (os.path.join(REPO, "profiles", n_) for n_ in ("calibration_r06.json", "calibration_r05.json"))
              if os.path.exists(p_)), os.path.join(REPO, "profiles", "calibration_r06.json"))
FP32_MATRIX_PEAK_TF = 157.3          # MI355X_MICROARCH.md: dense fp32 MFMA peak
BF16_MATRIX_PEAK_TF = 2500.0         # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (~2.5 PFLOP/s)
HBM_PEAK_GBS = 8000.0


def trunk_flops(batch):
    return 2.0 * batch * N_FILTER * N_FILTER * 9 * H * W


def load_mean_plies():
    if os.path.exists(CALIB):
        with open(CALIB) as f:
            c = json.load(f)
        return float(c["mean_plies_per_game"]), "profiles/%s (%d complete games counted in steady state, MI355X)" % (os.path.basename(CALIB), c["games"])
    return None, None


def load_counted():
    """The directly counted rate of the last `--count-games` run on this configuration (a committed measurement, not part
    of this run's timed region)."""
    if not os.path.exists(CALIB):
        return None
    with open(CALIB) as f:
        c = json.load(f)
    return {"games_per_s": c.get("games_per_s_counted"), "games": c.get("games"), "leaf_evals_per_s": c.get("leaf_evals_per_s"),
            "source": c.get("source")}


def _cpu_worker(budget_s):
    """One sequential self-play search loop on one core: oracle tree + oracle/net_ref.c batch-1 forward per playout."""
    from oracle.board_ref import RefBoard
    from oracle.mcts_ref import RefMCTS
    from oracle.net_ref_c import CNet
    prm = weights.init_params("resnet", H, W, 9, N_BLOCKS, N_FILTER, seed=0, style="bench")
    net = CNet(prm, H, W, 9, N_FILTER, N_BLOCKS, fast=True)    # the vectorised forward (checked against the plain loops in tests/test_oracle_net.py)
    b = RefBoard(W, H, N_IN_ROW)
    b.init_board()
    mcts = RefMCTS(net.policy_value_fn, c_puct=5, n_playout=N_PLAYOUT)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s:
        mcts.playout(b.clone())
        n += 1
    return n, time.perf_counter() - t0, net.isa


def cpu_baseline(mean_plies, cores, budget_s=15.0):
    """The sequential oracle (scalar tree, one batch-1 forward per playout) as `cores` independent self-play
    workers, one process per host core (child processes: this one has the GPU open), ~15 s each."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(budget_s)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=REPO) for _ in range(cores)]
    total_n, rates, isa = 0, [], None
    for p in procs:
        out, _ = p.communicate()
        try:
            r = json.loads(out.decode().strip().splitlines()[-1])
            total_n += r["n"]
            rates.append(r["n"] / r["dt"])
            isa = r.get("isa", isa)
        except (ValueError, IndexError, KeyError):
            pass
    if not rates:                      # no child came back: measure in-process on one core
        n, dt, isa = _cpu_worker(budget_s)
        total_n, rates = n, [n / dt]
    leaf_s = sum(rates)
    cpu = "?"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": leaf_s / (N_PLAYOUT * mean_plies), "unit": "games/s", "cores": len(rates), "kind": "port",
            "leaf_evals_per_s": leaf_s, "leaf_evals_per_s_per_core": leaf_s / len(rates), "cpu": cpu, "net_isa": isa,
            "sample": "%d sequential playouts in %d independent 15x15 self-play searches, one single-threaded process per core "
                      "(oracle tree + oracle/net_ref.c batch-1 forward per playout: vectorised direct convolution, %s), "
                      "%.0f s each; games/s = leaf-evals/s / (400 * mean plies).  Representative of the reference's own tree: "
                      "with the same net the imported reference MCTSPlayer runs 66.1 playouts/s per core and this oracle 68.2 "
                      "(same moves; profiles/r05_cpu_baseline_crosscheck.json, tools/cpu_baseline_crosscheck.py, build container)"
                      % (total_n, len(rates), isa, budget_s)}


def exchange_probe_world1(rows, timeout_s=75.0):
    """The N = 1 line's `exchange`: the same dist.measure_exchange as the N > 1 line runs inline, in a child process that
    forms a process group of one rank (RCCL on the GPU box) -- a child so that a backend that fails to come up costs
    this object, not the line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    # Bounded whatever the child does: its own session, killed as a group at the deadline, and ABANDONED if it does not
    # die (a child stuck inside the driver cannot be waited for) -- the line is worth more than this object
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--exchange-probe", str(int(rows))], env=env, cwd=REPO,
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout_s)
        obj = json.loads(out.decode().strip().splitlines()[-1])
        obj["process"] = "child process, process group of one rank"
        return obj
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, 9)
        except OSError:
            pass
        out = b""
        try:
            out, _ = p.communicate(timeout=5)
        except Exception:      # noqa: BLE001
            pass
        for ln in reversed((out or b"").decode(errors="replace").strip().splitlines()):      # the line may have been printed
            if ln.startswith("{"):                                                            # before the child got stuck
                try:
                    obj = json.loads(ln)
                    obj["process"] = "child process, process group of one rank (killed at the deadline after it had answered)"
                    return obj
                except ValueError:
                    pass
        return {"error": "the one-rank process group did not answer within %.0f s (child killed or abandoned)" % timeout_s}
    except Exception as e:            # noqa: BLE001 -- reported on the line
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def stem_roofline(device):
    """north_star target kernel: batched stem conv at 8192 x C_in x 15 x 15 (HBM-bound shape)."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from kernel_bench import synth
    out = {}
    n = 8192
    for c_in in (4, 9):
        prm = weights.init_params("resnet", H, W, c_in, 1, N_FILTER, seed=0, style="bench")
        net = PolicyValueNet(W, H, batch_size=n, n_blocks=1, n_filter=N_FILTER, model_params=prm, c_in=c_in,
                             device=device)
        _, planes = synth(n, W, c_in)
        net.forward_planes(planes)
        ms = net.conv_bench(0, n, iters=50, warmup=10)
        alg = n * (c_in * H * W + N_FILTER * H * W) * 4 + N_FILTER * c_in * 9 * 4
        traffic = None
        tpath = os.path.join(REPO, "profiles", "r05_stem_traffic.json")
        if os.path.exists(tpath):                      # PMC passes of rocprofv3 on tools/stem_profile.py (same launches)
            with open(tpath) as f:
                traffic = json.load(f).get("c_in_%d" % c_in, {}).get("traffic_bytes_per_launch")
        out["c_in_%d" % c_in] = {"bound": "hbm", "achieved": alg / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": alg / ms / 1e6 / HBM_PEAK_GBS, "traffic": traffic, "us_per_launch": ms * 1e3,
                                 "shape": "%dx%dx15x15 -> 128 ch" % (n, c_in), "algorithmic_bytes": alg,
                                 "tflops": 2.0 * n * c_in * 9 * N_FILTER * H * W / ms / 1e9}
        net.close()
    return out


ARITH_INFO = {
    # arith: (dtype text, kernel, executed matrix flops per board PAIR and launch, peak TF, flop text, numerics)
    "f32": ("f32 (exact fp32 products on the fp32 matrix pipe)", "trunk15_wino3_kernel<RESID>", 2 * 9216 * 2048.0, FP32_MATRIX_PEAK_TF,
            "9216 v_mfma_f32_16x16x4_f32 x 2048 flop per board",
            "tests/test_gpu_winograd_numerics.py::test_winograd_trunk_keeps_a_3x_margin_under_stress, profiles/r04_winograd_numerics.json"),
    "f16x2": ("f32 (split operands on the bf16/fp16 matrix pipe, fp32 accumulate; <=1e-4 vs float64 oracle)", "trunk15_wino3h_kernel<RESID>",
              36 * 4 * 16 * 2 * 32768.0, BF16_MATRIX_PEAK_TF,
              "36 positions x 4 groups of 32 output channels x 16 chunks x 2 v_mfma_f32_32x32x16_f16 (32 768 flop) per board pair: "
              "three fp16 products (+ lo.lo) per fp32 product",
              "tests/test_gpu_winograd_numerics.py::test_f16x2_split_trunk_is_fp32_accurate_under_stress, "
              "profiles/r06_winograd_numerics_f16x2.json"),
    "bf16x3": ("f32-accurate via 3 x bf16 split, fp32 accumulate (PolicyValueNet(trunk_arith='bf16x3'))", "trunk15_wino3b_kernel<RESID>",
               36 * 4 * 16 * 3 * 32768.0, BF16_MATRIX_PEAK_TF,
               "36 positions x 4 groups of 32 output channels x 16 chunks x 3 v_mfma_f32_32x32x16_bf16 (32 768 flop) per board pair: "
               "six bf16 products per fp32 product",
               "tests/test_gpu_winograd_numerics.py::test_bf16x3_split_trunk_is_fp32_accurate_under_stress, "
               "profiles/r04_winograd_numerics_bf16x3.json"),
}


def resolve_arith(arith):
    return "f16x2" if arith == "auto" else arith


def arith_line(arith, device, threads, G, pipeline, mean_plies, steps=120, warmup=30):
    """The SAME workload (BASELINE configs[2]) on another trunk arithmetic (PolicyValueNet(trunk_arith=...)): an EXTRA object
    beside the line's own `value`.  Own engine, own timed region (declared: `warmup` untimed steps + the 0.6 s pre-warm, then
    `steps` timed ones), trunk launches timed by HIP events on the engine stream."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    from alphapig_amd.selfplay import SelfPlayEngine
    prm = weights.init_params("resnet", H, W, 9, N_BLOCKS, N_FILTER, seed=0, style="bench")
    batch = (G + pipeline - 1) // pipeline
    net = PolicyValueNet(W, H, batch_size=batch, n_blocks=N_BLOCKS, n_filter=N_FILTER, model_params=prm, device=device,
                         trunk_arith=arith)
    eng = SelfPlayEngine(net, W, H, N_IN_ROW, n_games=G, n_playout=N_PLAYOUT, c_puct=5, temp=1.0, base_seed=20260000,
                         n_threads=threads, pipeline=pipeline)
    eng.run_steps(0)
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < 0.6:
        net.prewarm(batch, 8)
        net.sync()
    eng.run_steps(warmup)
    net.prewarm(batch, 4)
    p0 = eng.stats["leaf_evals"] + eng.terminal_playouts()
    l0 = eng.stats["leaf_evals"]
    net.set_profiling(max(1, (steps * pipeline) // 40))
    t0 = time.perf_counter()
    eng.run_steps(steps)
    net.sync()
    dt = time.perf_counter() - t0
    trunk_ms, trunk_cnt = net.kernel_time_ms("trunk")
    fwd_ms, fwd_cnt = net.kernel_time_ms("forward")
    net.set_profiling(False)
    playouts = eng.stats["leaf_evals"] + eng.terminal_playouts() - p0
    leafs = eng.stats["leaf_evals"] - l0
    overflows = net.trunk_overflows()
    eng.close()
    net.close()
    us = 1e3 * trunk_ms / max(trunk_cnt, 1)
    dtype, kernel, flops_pair, peak, flop_text, numerics = ARITH_INFO[arith]
    executed = (batch / 2.0) * flops_pair
    tf = executed / (us * 1e-6) / 1e12 if trunk_cnt else None
    return {"trunk_arith": arith, "dtype": dtype,
            "value": playouts / N_PLAYOUT / mean_plies / dt, "unit": "games/s", "leaf_evals_per_s": leafs / dt,
            "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
            "gpu_busy_frac": (fwd_ms * 1e-3 / dt) if fwd_cnt else None, "forwards_repeated_on_the_exact_kernel": overflows,
            "roofline": {"kernel": kernel, "bound": "mfma", "achieved": tf, "peak": peak,
                         "unit": "TFLOP/s", "frac": (tf / peak) if tf else None, "us_per_launch": us,
                         "launches": trunk_cnt, "boards_per_launch": batch, "flops_per_launch": executed,
                         "achieved_basis": "matrix flops the kernel executes (" + flop_text + ") / average launch duration (HIP events)",
                         "fp32_equivalent_tflops": (batch * 9216 * 2048.0 / (us * 1e-6) / 1e12) if trunk_cnt else None},
            "numerics": numerics}


def counted_child(arith, games, pipeline, window_s=60.0, warmup_max_s=260.0, timeout_s=400.0):
    """Games/s COUNTED inside this run: a fresh child process (`bench.py --count-games`) plays the same configuration with
    continuous refill until every slot has finished a game (or `warmup_max_s`), then counts finished games over `window_s`.
    Returns the child's JSON object, or {"error": ...} on a deadline (the line then falls back to the derived value)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--count-games", str(window_s), "--count-warmup-max", str(warmup_max_s),
           "--games", str(games), "--pipeline", str(pipeline), "--trunk-arith", arith, "--no-extras"]
    t0 = time.perf_counter()
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, start_new_session=True)
    except OSError as e:
        return {"error": "could not start the child: %s" % e}
    out = ""
    try:
        out, _ = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, 9)
        except OSError:
            pass
        try:
            out, _ = proc.communicate(timeout=10)
        except Exception:
            pass
        return {"error": "deadline of %.0f s passed" % timeout_s, "seconds": time.perf_counter() - t0}
    for ln in reversed(out.strip().splitlines()):
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                obj = json.loads(ln)
                obj["seconds_total"] = time.perf_counter() - t0
                return obj
            except ValueError:
                pass
    return {"error": "no JSON line from the child (exit code %s)" % proc.returncode, "seconds": time.perf_counter() - t0}


def train_step_line(device, steps=10, warmup=3):
    """SURVEY 8(f1) beside the hot path: the training step that consumes the games (policy_value_net_mxnet.py:282-299), 10
    blocks / 128 filters / 15x15, every operator a HIP kernel of this repository (alphapig_amd/train.py) -- at the reference's
    batch size (conf/train_config.yaml: 128) and at 512.  `ms_per_step`: host arrays in, (loss, entropy) out, as the
    reference's train_step signature has it; `ms_per_step_uploaded_batch`: on HipTrainer.upload's device copy, which is what
    the epochs of a policy_update pay.  An EXTRA object (tools/train_bench.py is the same measurement)."""
    import torch
    from alphapig_amd.train import HipTrainer
    rs = np.random.RandomState(0)
    prm = weights.init_params("resnet", H, W, 9, N_BLOCKS, N_FILTER, seed=0, style="bench")
    out = {"net": "10 blocks x 128 filters, 15x15, fp32", "steps": steps, "warmup": warmup}
    for B in (128, 512):
        states = (rs.rand(B, 9, H, W) > 0.7).astype(np.float32)
        pis = rs.dirichlet(np.ones(H * W), size=B).astype(np.float32)
        zs = rs.choice([-1.0, 1.0], size=B).astype(np.float32)
        tr = HipTrainer(prm, "resnet", n_blocks=N_BLOCKS, batch_size=B, device_index=device)
        for _ in range(warmup):
            tr.train_step(states, pis, zs, 1e-3)
        torch.cuda.synchronize(device)
        t = time.perf_counter()
        for _ in range(steps):
            tr.train_step(states, pis, zs, 1e-3)
        torch.cuda.synchronize(device)
        ms = 1e3 * (time.perf_counter() - t) / steps
        batch = tr.upload(states, pis, zs)
        tr.train_step(batch, None, None, 1e-3)
        torch.cuda.synchronize(device)
        t = time.perf_counter()
        for _ in range(steps):
            tr.train_step(batch, None, None, 1e-3)
        torch.cuda.synchronize(device)
        ms_up = 1e3 * (time.perf_counter() - t) / steps
        tr.close()
        out["batch_%d" % B] = {"ms_per_step": ms, "ms_per_step_uploaded_batch": ms_up}
    return out


def config2_line(device, steps=6000, warmup=800, trunk_arith="auto"):
    """BASELINE configs[1] (the metric's CPU-runnable sibling: 64 concurrent 8x8 games, 4 in a row, n_playout 200, the 6-conv
    net) through the same engine: leaf evaluations per second.  An EXTRA object (tests/config_table.py is the same measurement)."""
    from alphapig_amd.policy_value_net import LanedEvaluator, PolicyValueNet
    from alphapig_amd.selfplay import SelfPlayEngine
    prm = weights.init_params("simple", 8, 8, 9, N_BLOCKS, N_FILTER, seed=0, style="bench")
    # (round 6: "auto" = "f16x2" on 8x8 boards -- the five convolutions with >= 64 input channels on conv8h_kernel, split fp16
    # operands, fp32 accumulate: 256 -> 256 at 32 boards 23.4 -> 11.8 us; profiles/r06_config2.md)
    net = PolicyValueNet(8, 8, batch_size=32, n_blocks=N_BLOCKS, n_filter=N_FILTER, model_params=prm, net_kind="simple", device=device,
                         trunk_arith=trunk_arith)
    # one engine handle (own HIP stream) per pipeline group: the two groups' seven-launch forwards overlap on the GPU
    # (round 5: 337-353 k -> 402-415 k leaf evaluations per second, profiles/r05_config2.md); same games, bit for bit
    ev = LanedEvaluator.like(net, 2)
    eng = SelfPlayEngine(ev, 8, 8, 4, n_games=64, n_playout=200, temp=1.0, base_seed=77, pipeline=2, forced_opening=False)
    eng.run_steps(warmup)
    ev.sync()
    l0 = eng.stats["leaf_evals"]
    t = time.perf_counter()
    eng.run_steps(steps)
    ev.sync()
    dt = time.perf_counter() - t
    leafs = eng.stats["leaf_evals"] - l0
    arith, repeats = net.trunk_arith, ev.trunk_overflows()
    eng.close()
    ev.close()
    return {"workload": "64 concurrent 8x8 games, n_in_row 4, n_playout 200, simple 6-conv net, 32-board forwards, one evaluator lane "
                        "(HIP stream) per pipeline group", "steps": steps, "evaluator_lanes": 2, "trunk_arith": arith,
            "dtype": ARITH_INFO[arith][0].replace("bf16/fp16", "fp16") if arith != "f32" else ARITH_INFO[arith][0],
            "forwards_repeated_on_exact_kernel": repeats,
            "leaf_evals_per_s": leafs / dt, "ms_per_step": 1e3 * dt / steps}


def config2_both():
    """BASELINE configs[1] in the default arithmetic (8x8: split fp16 operands) and, beside it, on the exact-fp32 kernels: the
    same games, the second number on the same object as `leaf_evals_per_s_exact_f32`."""
    obj = config2_child()
    exact = config2_child(trunk_arith="f32")
    obj["leaf_evals_per_s_exact_f32"] = exact.get("leaf_evals_per_s") if isinstance(exact, dict) else None
    return obj


def config2_child(timeout_s=120.0, trunk_arith="auto"):
    """config2_line in a FRESH process: inside the bench process, after the other extras have created and destroyed half a
    dozen engines, the two lanes' streams no longer overlap (353 k instead of 385 - 415 k leaf evaluations/s, measured); a
    process that creates just these two streams gets two hardware queues.  Bounded like the exchange probe."""
    import subprocess
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--config2-worker", "--trunk-arith", trunk_arith], cwd=REPO,
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout_s)
        obj = json.loads(out.decode().strip().splitlines()[-1])
        obj["process"] = "child process (fresh HIP context)"
        return obj
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, 9)
        except OSError:
            pass
        try:
            p.communicate(timeout=5)
        except Exception:      # noqa: BLE001
            pass
        return {"error": "no answer within %.0f s" % timeout_s}
    except Exception as e:            # noqa: BLE001 -- reported on the line
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def _rank_log_dir(tag=None):
    """gpurun_out/ (travels back from the GPU box), one sub-directory per self-spawned job: two bench runs on one host
    must not truncate each other's rank logs, and the supervisor's silence watchdog sums the sizes of its OWN ranks' logs.
    APZ_RANK_LOG_DIR moves the parent directory (the test-suite points it at pytest's tmp_path: tests/conftest.py)."""
    d = os.environ.get("APZ_RANK_LOG_DIR") or os.path.join(REPO, "gpurun_out")
    if tag is not None:
        d = os.path.join(d, "ranks_%s" % tag)
    try:
        os.makedirs(d, exist_ok=True)
        return d
    except OSError:
        import tempfile
        return tempfile.mkdtemp(prefix="apz_ranks_")


def latency_probe(device):
    """The drop-in single-game API as the reference calls it (batch 1): one PolicyValueNet.policy_value_fn(board)
    (policy_value_net_mxnet.py:261-280) and one MCTSPlayer.get_action at n_playout = 400 (mcts_alphaZero.py:187-218) on the
    bench's 10-block net -- the small-batch kernels' line (csrc/trunk15_wino3s.h)."""
    from alphapig_amd.game import Board
    from alphapig_amd.mcts_alphaZero import MCTSPlayer
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", H, W, 9, N_BLOCKS, N_FILTER, seed=0, style="bench")
    net = PolicyValueNet(W, H, batch_size=16, n_blocks=N_BLOCKS, n_filter=N_FILTER, model_params=prm, device=device)
    b = Board(width=W, height=H, n_in_row=N_IN_ROW)
    b.init_board(0)
    for m in (112, 113, 97):
        b.do_move(m)
    for _ in range(50):
        net.policy_value_fn(b)
    t = time.perf_counter()
    for _ in range(300):
        net.policy_value_fn(b)
    leaf_ms = (time.perf_counter() - t) / 300 * 1e3
    p = MCTSPlayer(net.policy_value_fn, c_puct=5, n_playout=N_PLAYOUT, is_selfplay=0)
    t = time.perf_counter()
    p.get_action(b)
    move_s = time.perf_counter() - t
    net.close()
    return {"policy_value_fn_ms": leaf_ms, "get_action_n_playout_400_s": move_s,
            "what": "batch-1 calls of the reference API on the 10-block net, one board per forward (round 2: 1.0 ms / 0.43 s)"}


def spawn_ranks(n, argv, deadline_s=1800.0, silence_s=420.0, early_exit_grace_s=60.0):
    """`--gpus N` outside a launcher: start N fresh rank processes (this process has not touched the GPU and never
    does), pass rank 0's JSON line through, fail -- non-zero exit, every rank killed -- if
      * a rank exits non-zero,
      * a rank exits 0 while others are still running `early_exit_grace_s` later (they wait for it in a collective),
      * the whole job passes `deadline_s`, or
      * no rank has written a byte to its log (gpurun_out/rank<r>.log: stdout of ranks >= 1, stderr of all; ranks print
        a heartbeat line per phase and every 20 s of a long loop) for `silence_s` -- a hung RCCL rendezvous or a wedged GPU.
    Children are always fresh processes: nothing that has touched the GPU is ever re-executed."""
    import socket
    import subprocess
    # a port BELOW the kernel's ephemeral range (32768+): a port number the kernel handed out for bind(0) may be
    # taken by an outgoing connection of one of the ranks before rank 0 listens on it; two concurrent bench runs
    # on one host draw independent random candidates out of 12 000
    import random
    port = None
    for _ in range(64):
        cand = random.randint(20000, 32000)
        with socket.socket() as sk:
            try:
                sk.bind(("127.0.0.1", cand))
            except OSError:
                continue
        port = cand
        break
    if port is None:
        raise SystemExit("bench.py: no free rendezvous port in 20000..32000")
    import tempfile
    logdir = _rank_log_dir("%d_%d" % (port, os.getpid()))
    procs, logs = [], []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), APZ_BENCH_SELF_SPAWNED="1")
        lf = open(os.path.join(logdir, "rank%d.log" % r), "wb")
        logs.append(lf)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, cwd=REPO,
                                      stdout=out0 if r == 0 else lf, stderr=lf))
    t_start = time.time()
    failed = None                                     # (rank or None, reason)
    first_clean_exit = None
    last_bytes, last_change = -1, t_start
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.2)
        now = time.time()
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0):
                failed = (r, "rank %d exited with code %s" % (r, rc))
                break
            if rc == 0 and first_clean_exit is None:
                first_clean_exit = (r, now)
        if failed is not None:
            break
        if first_clean_exit is not None and now - first_clean_exit[1] > early_exit_grace_s:
            failed = (first_clean_exit[0], "rank %d exited 0 but the others are still running %.0f s later" % (
                first_clean_exit[0], early_exit_grace_s))
        total = 0
        for r in range(n):
            try:
                total += os.path.getsize(os.path.join(logdir, "rank%d.log" % r))
            except OSError:
                pass
        if total != last_bytes:
            last_bytes, last_change = total, now
        if failed is None and now - last_change > silence_s:
            failed = (None, "no rank wrote to its log for %.0f s (hung rendezvous / collective / GPU?)" % silence_s)
        if failed is None and now - t_start > deadline_s:
            failed = (None, "job deadline of %.0f s passed" % deadline_s)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    for lf in logs:
        lf.close()
    out0.seek(0)
    text = out0.read().decode()
    if failed is None:
        sys.stdout.write(text)
        sys.stdout.flush()
        return
    tails = []
    for r in range(n):
        try:
            with open(os.path.join(logdir, "rank%d.log" % r), "rb") as f:
                tails.append("--- rank %d log tail ---\n%s" % (r, f.read()[-600:].decode(errors="replace")))
        except OSError:
            pass
    sys.stderr.write("\n".join(tails) + "\n")
    raise SystemExit("bench.py: %s; stopped the other ranks (logs: %s/rank*.log)" % (failed[1], logdir))


def heartbeat(msg):
    """One line on stderr (a rank's log under the self-spawning supervisor, which treats a long silence as a hang)."""
    print("[bench rank %s %.1fs] %s" % (os.environ.get("RANK", "0"), time.perf_counter() - _T_PROC, msg), file=sys.stderr, flush=True)


_T_PROC = time.perf_counter()


def local_rank_world():
    """(rank, world) WITHIN this node: LOCAL_RANK / LOCAL_WORLD_SIZE as torchrun and spawn_ranks export them; a launcher
    that sets only RANK / WORLD_SIZE is taken to be single-node."""
    world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    return rank % max(world, 1), max(world, 1)


def pin_rank_cpus(rank, world):
    """Give LOCAL rank r of the `world` ranks on this node its own contiguous slice of the CPUs this job may use, BEFORE anything touches the GPU
    (threads created later -- OpenMP workers, HIP runtime, pipeline workers -- inherit it): eight ranks' tree threads
    otherwise migrate across both sockets.  Contiguous slices keep a rank inside one NUMA node on the usual layouts
    (GPUs 0-3 on socket 0, 4-7 on socket 1).  APZ_BENCH_NO_AFFINITY=1 switches it off.  -> CPUs in the slice or None."""
    if world <= 1 or os.environ.get("APZ_BENCH_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    per = len(cpus) // world
    if per < 1:
        return None
    mine = cpus[rank * per:(rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return len(mine)


def host_cpu_share():
    """CPUs this process may use: affinity mask, capped by the cgroup CPU quota (a 1-GPU box shows 256 CPUs and
    grants 16 cores' worth of time; all ranks of a node share it)."""
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            ncpu = min(ncpu, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return ncpu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--games", type=int, default=GAMES_PER_GPU, help="concurrent games per GPU")
    ap.add_argument("--pipeline", type=int, default=2)
    ap.add_argument("--full-games", type=int, default=0, help="play this many COMPLETE games per GPU "
                    "instead of timing --steps (measures games/s and mean plies directly; minutes)")
    ap.add_argument("--count-games", type=float, default=0.0, metavar="SECONDS",
                    help="count finished games over a steady-state window of this many seconds (continuous refill; the "
                         "window opens once every slot has finished a game or after --count-warmup-max seconds)")
    ap.add_argument("--count-warmup-max", type=float, default=420.0)
    ap.add_argument("--no-extras", action="store_true", help="skip roofline_stem, latency, trunk_exact_f32, train_step, config2, cpu_baseline and the counted leg")
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)   # child of cpu_baseline()
    ap.add_argument("--exchange-probe", type=int, default=0, metavar="ROWS", help=argparse.SUPPRESS)   # child of exchange_probe_world1()
    ap.add_argument("--config2-worker", action="store_true", help=argparse.SUPPRESS)                   # child of config2_child()
    ap.add_argument("--profile-every", type=int, default=16, help="HIP-event-time every k-th forward in the timed region")
    ap.add_argument("--profile-samples", type=int, default=20, help="... or more often, for at least this many timed forwards")
    ap.add_argument("--mean-plies", type=float, default=None, help="debug override of the calibrated mean plies/game")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 REHEARSAL on a one-GPU box: every rank uses cuda:0 and the collectives run on gloo (RCCL refuses two "
                         "ranks on one device).  Exercises the real evaluator under the multi-rank path; the line is marked "
                         "invalid: the ranks share one GPU, so it is NOT a scaling measurement")
    ap.add_argument("--deadline-s", type=float, default=1800.0, help="self-spawned ranks: kill everything after this many seconds")
    ap.add_argument("--rank-silence-s", type=float, default=420.0,
                    help="self-spawned ranks: kill everything when no rank has written to its log for this long")
    ap.add_argument("--early-exit-grace-s", type=float, default=60.0, help=argparse.SUPPRESS)
    ap.add_argument("--prewarm-s", type=float, default=0.6,
                    help="declared UNTIMED GPU-only pre-warm before the W warm-up steps: dummy forwards of empty boards "
                         "(not engine steps) for this many seconds, plus 4 dummy forwards queued right before the closing "
                         "synchronisation of the warm-up so that the timed region starts on a GPU at its running clocks; 0 = off")
    ap.add_argument("--trunk-arith", default="auto", choices=["auto", "f32", "f16x2", "bf16x3"],
                    help="arithmetic of the trunk convolutions on batches of more than 32 boards (PolicyValueNet(trunk_arith=...)); "
                         "auto = f16x2")
    ap.add_argument("--only-arith", default=None, choices=["f32", "f16x2", "bf16x3"],
                    help="run ONLY the extra engine run on this trunk arithmetic and print its object: the command a rocprofv3 "
                         "trace of one kernel is taken of")
    ap.add_argument("--no-count", action="store_true", help="skip the counted games/s leg (a child process, ~4 minutes)")
    ap.add_argument("--count-window", type=float, default=60.0, help="window of the counted leg, seconds")
    ap.add_argument("--plumbing-test", action="store_true",
                    help="CPU self-test of the multi-rank plumbing (gloo, stand-in evaluator from tests/fakenet.py, "
                         "8 games per rank); the JSON line is marked invalid and is NOT a measurement")
    args = ap.parse_args()
    if args.cpu_worker > 0:
        n_, dt_, isa_ = _cpu_worker(args.cpu_worker)
        print(json.dumps({"n": n_, "dt": dt_, "isa": isa_}))
        return

    if args.config2_worker:
        print(json.dumps(config2_line(0, trunk_arith=args.trunk_arith)))
        return
    if args.exchange_probe > 0:
        dist.init(force=True)                        # a process group of THIS rank alone: RCCL when there is a GPU, else gloo
        print(json.dumps(dist.measure_exchange(args.exchange_probe)), flush=True)
        try:
            dist.shutdown()
        finally:
            os._exit(0)                              # a backend that hangs in its teardown must not cost the printed line
    if args.only_arith:
        mean_plies, _ = load_mean_plies()
        obj = arith_line(args.only_arith, 0, max(1, min(int(os.environ.get("APZ_HOST_THREADS", "16")), host_cpu_share())), args.games,
                         args.pipeline, mean_plies, steps=args.steps if args.steps != 1200 else 120,
                         warmup=args.warmup if args.warmup != 100 else 30)
        print(json.dumps(obj))
        return
    arith = resolve_arith(args.trunk_arith)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:], deadline_s=args.deadline_s, silence_s=args.rank_silence_s,
                           early_exit_grace_s=args.early_exit_grace_s)     # before anything touches the GPU
    if os.environ.get("APZ_BENCH_TEST_FAIL_RANK") == os.environ.get("RANK", "0"):
        raise SystemExit(3)                                  # tests/test_dist_gloo.py: a rank that dies
    if os.environ.get("APZ_BENCH_TEST_HANG_RANK") == os.environ.get("RANK", "0"):
        time.sleep(3600)                                     # tests/test_dist_gloo.py: a rank that hangs before the rendezvous
    if os.environ.get("APZ_BENCH_TEST_EARLY_EXIT_RANK") == os.environ.get("RANK", "0"):
        raise SystemExit(0)                                  # ... and one that leaves early with exit code 0
    ncpu = host_cpu_share()                                  # the JOB's CPU share (before this rank narrows its own mask)
    lrank, lworld = local_rank_world()
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(lworld))   # the tree pool divides its pre-touch limit by it (host_tree.cpp)
    pinned = pin_rank_cpus(lrank, lworld)                    # before any GPU call
    heartbeat("starting (cpu slice: %s)" % pinned)
    rank, world, local = dist.init(backend="gloo" if (args.plumbing_test or args.rehearse_on_one_gpu) else None)
    if args.rehearse_on_one_gpu:
        local = 0
    heartbeat("process group up: rank %d of %d" % (rank, world))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    from alphapig_amd.policy_value_net import PolicyValueNet
    from alphapig_amd.selfplay import SelfPlayEngine

    # host threads for the tree pool: this node's CPU share split evenly between its ranks
    threads = max(1, min(int(os.environ.get("APZ_HOST_THREADS", "16")), ncpu // max(lworld, 1)))
    if pinned:
        threads = max(1, min(threads, pinned))
    G = args.games
    if args.plumbing_test:
        class _StandIn(object):      # counts like the real evaluator; numbers come from tests/fakenet.py
            def __init__(self):
                sys.path.insert(0, os.path.join(REPO, "tests"))
                from fakenet import fake_policy_value_batch
                self.fn, self.pool = fake_policy_value_batch, None

            def evaluate_codes(self, codes):
                return self.fn(self.pool.codes_to_planes(codes, 9))

            def sync(self):
                pass

            def set_profiling(self, on):
                pass

            def kernel_time_ms(self, k):
                return 0.0, 0

            def close(self):
                pass
        G = 8
        net = _StandIn()
    else:
        prm = weights.init_params("resnet", H, W, 9, N_BLOCKS, N_FILTER, seed=0, style="bench")
        # one evaluator, one HIP stream; the pipeline groups queue their batches on it back to back
        net = PolicyValueNet(W, H, batch_size=(G + args.pipeline - 1) // args.pipeline, n_blocks=N_BLOCKS,
                             n_filter=N_FILTER, model_params=prm, device=local, trunk_arith=arith)
    lanes = [net]
    eng = SelfPlayEngine(net, W, H, N_IN_ROW, n_games=G, n_playout=N_PLAYOUT, c_puct=5, temp=1.0, base_seed=20260000,
                         n_threads=threads, pipeline=args.pipeline, index_offset=rank, index_stride=world)

    if args.plumbing_test:
        net.pool = eng.pool

    def playouts_done():
        return eng.stats["leaf_evals"] + eng.terminal_playouts()

    mean_plies, plies_src = load_mean_plies()
    if args.mean_plies:
        mean_plies, plies_src = args.mean_plies, "--mean-plies override (debug)"
    if args.full_games:
        dist.barrier()
        t0 = time.perf_counter()
        def progress(e):
            print("[full-games] %.0fs games %d/%d moves %d leaf-evals %d" % (
                time.perf_counter() - t0, e.stats["games"], args.full_games, e.stats["moves"],
                e.stats["leaf_evals"]), file=sys.stderr, flush=True)
        eps = eng.play_games(args.full_games, progress=progress)
        dist.barrier()
        dt = dist.all_reduce_max(time.perf_counter() - t0)
        games = dist.all_reduce_sum(eng.stats["games"])
        plies = dist.all_reduce_sum(eng.stats["plies"])
        leafs = dist.all_reduce_sum(eng.stats["leaf_evals"])
        # note: games still in flight when the target was reached are not counted (conservative)
        if rank == 0:
            res = {"mode": "full-games", "games": int(games), "mean_plies_per_game": plies / games,
                   "seconds": dt, "games_per_s_completed_only": games / dt, "leaf_evals_per_s": leafs / dt,
                   "leaf_evals": int(leafs), "n_gpus": world, "games_per_gpu_concurrent": G,
                   "winners": {str(k): int(sum(1 for e in eps if e.winner == k)) for k in (-1, 1, 2)}}
            print(json.dumps(res))
        eng.close()
        for ln in lanes:
            ln.close()
        return
    if args.count_games > 0:
        # Steady state, counted: slots are refilled the moment a game ends; once every slot has finished at least one
        # game (so the mix of game ages is the stationary one) finished games are counted over a fixed window.
        t_start = time.perf_counter()
        last = t_start
        all_done_at = None
        while True:
            eng.run_steps(64)
            now = time.perf_counter()
            frac = float(np.mean(eng.slot_games > 0))
            if frac == 1.0 or now - t_start > args.count_warmup_max:
                all_done_at = now - t_start
                break
            if now - last > 20.0:
                print("[count-games] warm-up %.0fs: %.1f%% of the slots have finished a game, %d games" %
                      (now - t_start, 100 * frac, eng.stats["games"]), file=sys.stderr, flush=True)
                last = now
        dist.barrier()
        g0, p0_, l0_ = eng.stats["games"], eng.stats["plies"], eng.stats["leaf_evals"]
        t0 = time.perf_counter()
        last = t0
        while time.perf_counter() - t0 < args.count_games:
            eng.run_steps(64)
            if time.perf_counter() - last > 20.0:
                print("[count-games] window %.0fs: %d games" % (time.perf_counter() - t0, eng.stats["games"] - g0),
                      file=sys.stderr, flush=True)
                last = time.perf_counter()
        for ln in lanes:
            ln.sync()
        dt = dist.all_reduce_max(time.perf_counter() - t0)
        games = dist.all_reduce_sum(eng.stats["games"] - g0)
        plies = dist.all_reduce_sum(eng.stats["plies"] - p0_)
        leafs = dist.all_reduce_sum(eng.stats["leaf_evals"] - l0_)
        if rank == 0:
            mp = plies / max(games, 1)
            print(json.dumps({"mode": "count-games", "games_in_window": int(games), "window_s": dt,
                              "games_per_s_counted": games / dt, "mean_plies_per_game": mp,
                              "leaf_evals_per_s": leafs / dt,
                              "games_per_s_derived": leafs / dt / (N_PLAYOUT * mp) if games else None,
                              "warmup_s": all_done_at, "slots_with_a_finished_game": float(np.mean(eng.slot_games > 0)),
                              "n_gpus": world, "games_per_gpu_concurrent": G, "host_threads_per_rank": threads,
                              "trunk_arith": arith,
                              "forwards_repeated_on_the_exact_kernel": sum(ln.trunk_overflows() for ln in lanes)}))
        eng.close()
        for ln in lanes:
            ln.close()
        return
    if mean_plies is None:
        raise SystemExit("profiles/calibration_r0x.json missing: run `python bench.py --count-games 240` once")

    import gc
    gc.collect()
    gc.freeze()                                      # no cyclic-GC pauses inside the timed region ...
    gc.disable()                                     # ... and no collection BETWEEN warm-up and timed region: tens of ms of idle GPU
    # HIP-event timing of the trunk launches: every k-th forward of the timed region, k small enough for >= 20 samples
    # (the driver's 20-step runs have 40 forwards: three samples, the first of them right behind the idle GPU of the
    # synchronisation below, are not an average); every forward as a whole is bracketed too (-> gpu_busy_frac)
    prof_stride = max(1, min(args.profile_every, (args.steps * args.pipeline) // max(1, args.profile_samples)))
    batch = G // args.pipeline
    # Declared, untimed, GPU-only pre-warm (profiles/r03_driver_window.md): a GPU that has just been idle runs the trunk
    # kernel at 118-127 us per launch and needs ~40 forwards (80 ms) to reach its running 98 us, and the first few
    # hundred launches / timing events of a process pay the HIP runtime's pool growth (ms-long host stalls).  With
    # W = 5 both landed inside the driver's 0.1 s window.  These are dummy forwards of empty boards on the evaluator's
    # stream -- no engine step, no playout is done here; `steps` and `warmup` keep their meaning.
    prewarm = {"seconds": 0.0, "dummy_forwards": 0, "keep_warm_forwards": 0, "boards_per_forward": batch,
               "what": "untimed GPU-only dummy forwards of empty boards before the warm-up steps (+ a few queued right before the "
                       "synchronisation that closes the warm-up): clock ramp and HIP runtime pool growth; no engine steps"}
    eng.run_steps(0)                                 # seeds the G games (tens of ms of host work) BEFORE the GPU is warmed
    if args.prewarm_s > 0 and not args.plumbing_test:
        t_pw = time.perf_counter()
        for ln in lanes:
            ln.set_profiling(prof_stride)            # the event path is part of what has to be warm
        while time.perf_counter() - t_pw < args.prewarm_s:
            for ln in lanes:
                ln.prewarm(batch, 8)
            for ln in lanes:
                ln.sync()
            prewarm["dummy_forwards"] += 8
        prewarm["seconds"] = time.perf_counter() - t_pw
    heartbeat("pre-warm done (%d dummy forwards)" % prewarm["dummy_forwards"])
    eng.run_steps(args.warmup)                       # W untimed warm-up steps
    heartbeat("warm-up steps done")
    if args.prewarm_s > 0 and not args.plumbing_test:
        for ln in lanes:
            ln.prewarm(batch, 4)                     # keeps the GPU busy across the bookkeeping below (not waited for here)
        prewarm["keep_warm_forwards"] = 4
    p0, l0 = playouts_done(), eng.stats["leaf_evals"]
    host0, eval0 = eng.timers["host_s"], eng.timers["eval_s"]
    eng.step_times = []
    dist.barrier()
    for ln in lanes:
        ln.set_profiling(prof_stride)                # stream synchronize (== torch.cuda.synchronize() for the engine stream) + counters reset
    t0 = time.perf_counter()
    eng.run_steps(args.steps)                        # exactly K timed steps
    # the round's exchange: all-gather of the tuples of games that finished inside the window
    done = [e for e in eng.finished]
    if world > 1:
        if done:
            codes = np.concatenate([e.codes for e in done])
            pis = np.concatenate([e.pis for e in done]).astype(np.float32)
            zs = np.concatenate([e.zs for e in done]).astype(np.float32)
        else:
            codes = np.zeros((0, eng.pool.code_stride), np.uint8)
            pis = np.zeros((0, H * W), np.float32)
            zs = np.zeros(0, np.float32)
        dist.all_gather_tuples(codes, pis, zs)
    for ln in lanes:
        ln.sync()
    dist.barrier()
    dt_local = time.perf_counter() - t0
    heartbeat("timed region done: %.3f s" % dt_local)
    dt = dist.all_reduce_max(dt_local)
    ranks_seen = int(round(dist.all_reduce_sum(1)))
    if ranks_seen != world:
        raise SystemExit("bench.py: the all-reduce saw %d ranks, WORLD_SIZE is %d" % (ranks_seen, world))
    per_rank_rate = dist.all_gather_floats((eng.stats["leaf_evals"] - l0) / dt_local)     # a straggler is invisible under the MAX
    rank_devices = [int(v) for v in dist.all_gather_floats(local)]                        # device ordinal of every rank
    pg = dist.group_info()
    # the round's exchange WITH a payload (outside the timed region): what a full round of G games per rank hands over
    exch_rows = int(round(G * mean_plies))
    exchange = dist.measure_exchange(exch_rows, code_stride=eng.pool.code_stride, hw=H * W) if world > 1 else None
    playouts = dist.all_reduce_sum(playouts_done() - p0)
    leafs = dist.all_reduce_sum(eng.stats["leaf_evals"] - l0)
    trunk_ms = trunk_cnt = fwd_ms = fwd_cnt = 0
    for ln in lanes:
        ms_, cnt_ = ln.kernel_time_ms("trunk")
        trunk_ms, trunk_cnt = trunk_ms + ms_, trunk_cnt + cnt_
        ms_, cnt_ = ln.kernel_time_ms("forward")
        fwd_ms, fwd_cnt = fwd_ms + ms_, fwd_cnt + cnt_
        ln.set_profiling(False)
    if rank != 0:
        eng.close()
        for ln in lanes:
            ln.close()
        return

    games_per_s = playouts / N_PLAYOUT / mean_plies / dt
    st_ms = sorted(1e3 * x for x in getattr(eng, "step_times", []))
    tk = {"f32": "wino3", "f16x2": "wino3h", "bf16x3": "wino3b"}[arith]      # the trunk kernel of this run's arithmetic
    # HBM / fabric traffic per launch of the dominant kernel: PMC passes of rocprofv3 on this same command
    # (cannot be collected from inside the process), committed under profiles/
    traffic, traffic_src = None, None
    for tname in ("r06_trunk_traffic.json", "r05_trunk_traffic.json", "r04_trunk_traffic.json", "r03_trunk_traffic.json", "r02_trunk_traffic.json"):
        tpath = os.path.join(REPO, "profiles", tname)
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("boards_per_launch") == batch and tj.get("kernel", "ring") == tk:
                traffic, traffic_src = tj["traffic_bytes_per_launch"]["mean"], "profiles/" + tname
                break
    trunk_avg_ms = trunk_ms / max(trunk_cnt, 1)
    dtype_text, kshort, flops_pair, peak_tf, flop_text, numerics_src = ARITH_INFO[arith]
    kernel_name = {"f32": "trunk15_wino3_kernel<RESID> (fused F(4x4,3x3) Winograd on fp32 MFMA, single pass: two boards x 64 output channels per work item)",
                   "f16x2": "trunk15_wino3h_kernel<RESID> (fused F(4x4,3x3) Winograd, every fp32 operand as two fp16 terms on the fp16 matrix pipe, "
                            "fp32 accumulation: two boards x 64 output channels per work item)",
                   "bf16x3": "trunk15_wino3b_kernel<RESID> (fused F(4x4,3x3) Winograd, three bf16 terms per operand)"}[arith]
    # Matrix flops the kernel really issues (ARITH_INFO).  `achieved` / `frac` are on THIS basis (a fraction of the peak of the
    # pipe the kernel runs on, <= 1); the rate of the layer's definition (direct-convolution flops / time) and the fp32
    # products per second (what the exact kernel's 9.66 GFLOP per launch would be) are kept beside it.
    executed = (batch / 2.0) * flops_pair
    executed_tf = executed / (trunk_avg_ms * 1e-3) / 1e12 if trunk_cnt else None
    direct_tf = trunk_flops(batch) / (trunk_avg_ms * 1e-3) / 1e12 if trunk_cnt else None
    fp32_equiv_tf = batch * 9216 * 2048.0 / (trunk_avg_ms * 1e-3) / 1e12 if trunk_cnt else None
    # algorithmic bytes of one launch (direct-convolution minimum): input + output planes (+ residual on
    # every second launch) in the rows16 layout + the layer's weights once
    alg_bytes = batch * N_FILTER * 960.0 * 2.5 + 9 * N_FILTER * N_FILTER * 4.0
    arena_gb = eng.pool.arena_bytes() / 1e9 if hasattr(eng.pool, "arena_bytes") else None
    line = {
        "metric": "self-play games/sec (15x15, n_playout=400)", "value": games_per_s, "unit": "games/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_text, "data": "synthetic",
        "trunk_arith": arith, "forwards_repeated_on_the_exact_kernel": sum(ln.trunk_overflows() for ln in lanes) if not args.plumbing_test else 0,
        "config": {"workload": "BASELINE configs[2]: 15x15, 5-in-row, n_playout=400, c_puct=5, temp=1.0, "
                               "10-block/128-filter residual net, %d concurrent games per GPU" % G,
                   "games_per_gpu": G, "leaf_batch": batch, "pipeline": args.pipeline, "host_threads": threads,
                   "mean_plies_per_game": mean_plies, "mean_plies_source": plies_src, "weights": "random init seed 0"},
        "parity_pin": "tree / board / self-play / augmentation: pinned to the reference's own outputs (tests/golden); network: "
                      "structure only (the reference's graph JSON) -- the 1e-4 is against this repo's float64 oracle, NOT "
                      "against MXNet (absent); see DESIGN.md section 2",
        "ranks_seen": ranks_seen, "process_group": pg, "rank_devices": rank_devices,
        "host_threads_per_rank": threads, "host_cpu_share": ncpu,
        "per_rank_leaf_evals_per_s": per_rank_rate, "cpus_pinned_per_rank": pinned,
        "warnings": (["host_threads_per_rank = %d < 3: the tree pool needs about three host threads per rank to keep one GPU "
                      "busy (DESIGN section 5); this run is host-bound" % threads] if threads < 3 else []),
        "tree_arena_gb_per_rank": arena_gb,
        "launcher": "self-spawned" if os.environ.get("APZ_BENCH_SELF_SPAWNED") else
                    ("torchrun" if "TORCHELASTIC_RUN_ID" in os.environ or world > 1 else "single process"),
        "leaf_evals_per_s": leafs / dt,
        "playouts_per_s": playouts / dt,
        "value_basis": "derived: playouts of the timed region / n_playout / mean plies per game (config.mean_plies_per_game)",
        "value_derived": games_per_s,
        "counted_steady_state": dict(load_counted() or {}, note="COMMITTED builder-run measurement (bench.py --count-games), echoed "
                                     "for context; NOT measured by this run (this run's own count: `counted`)"),
        "host_tree_s": eng.timers["host_s"] - host0, "evaluator_s": eng.timers["eval_s"] - eval0, "wall_s": dt,
        # GPU time of the timed region: every forward (stem .. value head, kernels back to back) bracketed by one HIP
        # event pair on the engine stream; busy fraction = their sum / wall clock of the timed region (rank 0)
        "gpu_busy_frac": (fwd_ms * 1e-3 / dt_local) if fwd_cnt else None,
        "kernel_ms_per_step": (fwd_ms / args.steps) if fwd_cnt else None,
        "forwards_timed": fwd_cnt,
        "gpu_bound": bool(fwd_cnt and fwd_ms * 1e-3 / dt_local >= 0.95),
        "step_ms": ({"median": st_ms[len(st_ms) // 2], "max": st_ms[-1], "min": st_ms[0]} if st_ms else None),
        "prewarm": prewarm,
        "roofline": {"kernel": kernel_name + ": trunk 128->128 3x3 conv + folded BN (+residual) + ReLU; 20 launches per forward",
                     "bound": "mfma", "achieved": executed_tf, "peak": peak_tf, "unit": "TFLOP/s",
                     "frac": (executed_tf / peak_tf) if executed_tf else None,
                     "achieved_basis": "matrix flops the kernel executes (" + flop_text + ") / "
                                       "average launch duration (HIP events on the engine stream inside the timed region)",
                     "flops_per_launch": executed,
                     "fp32_products_equivalent": {"tflops": fp32_equiv_tf, "frac_of_fp32_matrix_peak": (fp32_equiv_tf / FP32_MATRIX_PEAK_TF) if fp32_equiv_tf else None,
                                                  "note": "the exact-fp32 kernel's 9.66 GFLOP per 512 boards / this kernel's time: what the "
                                                          "fp32 pipe would have to sustain to match it (peak 157.3)"},
                     "hbm_view": {"algorithmic_bytes_per_launch": alg_bytes, "achieved_gbs": (alg_bytes / (trunk_avg_ms * 1e-3) / 1e9) if trunk_cnt else None,
                                  "peak_gbs": HBM_PEAK_GBS, "frac": (alg_bytes / (trunk_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if trunk_cnt else None,
                                  "note": "direct-convolution minimum bytes (input + output, residual on every second launch, weights once) / "
                                          "the same time: neither the matrix pipe nor HBM bounds this kernel -- DESIGN section 9"},
                     "numerics": numerics_src,
                     "direct_conv_equivalent": {"tflops": direct_tf, "flops_per_launch": trunk_flops(batch),
                                                "note": "SURVEY 8(d) algorithmic flops of the layer (2*9*128*128*225 per board) / the "
                                                        "same time; exceeds the peak because Winograd executes 3.5x fewer flops"},
                     "traffic": traffic,
                     "traffic_unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE from separate rocprofv3 --pmc passes, %s; "
                                     "direct-convolution minimum 158 MB, see there)" % traffic_src,
                     "us_per_launch": trunk_avg_ms * 1e3, "launches": trunk_cnt, "boards_per_launch": batch},
    }
    eng.close()
    for ln in lanes:
        ln.close()
    if args.plumbing_test:
        line["valid"] = False
        line["data"] = "plumbing-test (CPU stand-in evaluator, 8 games per rank): NOT a measurement"
    if args.rehearse_on_one_gpu:
        line["valid"] = False
        line["data"] = "REHEARSAL: %d ranks share ONE GPU, collectives on gloo: plumbing of the N > 1 path with the real evaluator, NOT a scaling measurement" % world
    if exchange is not None:
        line["exchange"] = exchange
    if world > 1:
        prev = None
        for name in ("bench_r05.json", "bench_r04.json"):
            path = os.path.join(REPO, "profiles", name)
            if os.path.exists(path):
                try:
                    with open(path) as f:
                        cb = json.load(f).get("cpu_baseline")
                    if cb:
                        prev = dict(cb, source="profiles/" + name)
                        break
                except (ValueError, OSError):
                    pass
        line["cpu_baseline"] = dict(prev or {}, note="timed on rank 0 of the N = 1 run only (the host cores are busy feeding N GPUs "
                                    "here); this is the committed N = 1 measurement, not part of this run")
    if not args.no_extras and world == 1 and not args.plumbing_test:
        for key, fn in (("roofline_stem", lambda: stem_roofline(local)), ("latency", lambda: latency_probe(local)),
                        ("trunk_exact_f32" if arith != "f32" else "trunk_f16x2",
                         lambda: arith_line("f32" if arith != "f32" else "f16x2", local, threads, G, args.pipeline, mean_plies)),
                        ("train_step", lambda: train_step_line(local)), ("config2", lambda: config2_both()),
                        ("cpu_baseline", lambda: cpu_baseline(mean_plies, cores=max(1, min(16, ncpu))))):
            heartbeat("extra: %s" % key)
            line[key] = fn()
        heartbeat("extra: exchange probe (child process, one-rank process group)")
        line["exchange"] = exchange_probe_world1(exch_rows)     # bounded, and nothing else depends on it
        if arith != "f32" and isinstance(line.get("trunk_exact_f32"), dict):
            line["value_exact_f32"] = line["trunk_exact_f32"].get("value")      # the same workload on the exact-fp32 trunk kernel
        if not args.no_count:
            heartbeat("counted leg: child process, continuous refill until every slot has finished a game, then a %.0f s window" % args.count_window)
            cnt = counted_child(arith, G, args.pipeline, window_s=args.count_window)
            line["counted"] = cnt
            if cnt.get("games_per_s_counted") and cnt.get("slots_with_a_finished_game", 0) >= 0.999:
                line["value"] = cnt["games_per_s_counted"]
                line["value_basis"] = ("COUNTED in this run: finished games / seconds over a %.0f s steady-state window of a child process "
                                       "of this command (continuous refill, opened once every slot had finished a game); mean plies per "
                                       "game of the window in counted.mean_plies_per_game; the rate derived from the timed region is "
                                       "value_derived" % cnt.get("window_s", args.count_window))
                line["counted_steady_state"]["note"] = "COMMITTED builder-run measurement, echoed for context; this run's own count is in `counted`"
            else:
                line["value_basis"] += " -- the counted leg did not complete (%s): value stays derived" % (cnt.get("error") or "window opened before every slot had finished a game")
        heartbeat("extras done")
    print(json.dumps(line))


if __name__ == "__main__":
    try:
        main()
    finally:
        if sys.exc_info()[0] is None:
            dist.shutdown()
