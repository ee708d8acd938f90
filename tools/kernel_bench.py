#!/usr/bin/env python3
"""Kernel micro-benchmarks on the GPU box (SURVEY.md 8d synthetic inputs): per-layer conv
timings via apz_conv3x3_bench (HIP events on the engine stream) and whole-forward timings."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

from alphapig_amd import weights  # noqa: E402
from alphapig_amd.policy_value_net import PolicyValueNet  # noqa: E402
from alphapig_amd.treepool import TreePool  # noqa: E402


def synth(n, w, c, seed=1234):
    rs = np.random.RandomState(seed)
    pool = TreePool(w, w, 5 if w >= 15 else 4, n_games=1, n_playout=1)
    codes = np.zeros((n, pool.code_stride), dtype=np.uint8)
    for i in range(n):
        k = int(rs.randint(0, min(81, w * w)))
        cells = rs.permutation(w * w)[:k]
        pool.set_position(0, cells, [1 + (j % 2) for j in range(k)], 1 + (k % 2))
        codes[i] = pool.codes(0)
    return codes, pool.codes_to_planes(codes, c)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-trunk", type=int, default=1024)
    ap.add_argument("--n-stem", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    out = {}
    H = W = 15
    # ---- stem conv at the north_star shape (C_in = 4) and the real shape (C_in = 9)
    for c_in in (4, 9):
        prm = weights.init_params("resnet", H, W, c_in, 1, 128, seed=0, style="bench")
        net = PolicyValueNet(W, H, batch_size=args.n_stem, n_blocks=1, n_filter=128, model_params=prm, c_in=c_in)
        _, planes = synth(args.n_stem, W, c_in)
        net.forward_planes(planes)
        ms = net.conv_bench(0, args.n_stem, iters=args.iters, warmup=10)
        alg = args.n_stem * (c_in * 225 + 128 * 225) * 4 + 128 * c_in * 9 * 4
        flop = 2.0 * args.n_stem * c_in * 9 * 128 * 225
        out["stem_c%d" % c_in] = dict(n=args.n_stem, ms=ms, alg_bytes=alg, gbps=alg / ms / 1e6,
                                      hbm_frac=alg / ms / 1e6 / 8000.0, tflops=flop / ms / 1e9)
        print("stem C_in=%d n=%d: %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)  %.1f TF" %
              (c_in, args.n_stem, ms * 1e3, alg / ms / 1e6, 100 * alg / ms / 1e6 / 8000.0, flop / ms / 1e9), flush=True)
        net.close()
    # ---- trunk conv + whole net at n = n_trunk
    prm = weights.init_params("resnet", H, W, 9, 10, 128, seed=0, style="bench")
    n = args.n_trunk
    net = PolicyValueNet(W, H, batch_size=n, n_blocks=10, n_filter=128, model_params=prm)
    codes, planes = synth(n, W, 9)
    net.forward_planes(planes)
    for layer in (1, 2):
        ms = net.conv_bench(layer, n, iters=args.iters, warmup=10)
        flop = 2.0 * n * 128 * 128 * 9 * 225
        out["trunk_l%d" % layer] = dict(n=n, ms=ms, tflops=flop / ms / 1e9, frac_fp32_peak=flop / ms / 1e9 / 157.3)
        print("trunk layer %d n=%d: %.1f us  %.1f TF (%.1f%% of 157.3)" % (layer, n, ms * 1e3, flop / ms / 1e9,
                                                                           100 * flop / ms / 1e9 / 157.3), flush=True)
    for label, fn, arg in (("planes", net.forward_planes, planes), ("codes", net.evaluate_codes, codes)):
        fn(arg)
        t = time.time()
        reps = 10
        for _ in range(reps):
            fn(arg)
        dt = (time.time() - t) / reps
        out["forward_" + label] = dict(n=n, ms=dt * 1e3, leaf_evals_per_s=n / dt)
        print("forward(%s) n=%d: %.2f ms -> %.0f leaf-evals/s" % (label, n, dt * 1e3, n / dt), flush=True)
    net.set_profiling(True)
    for _ in range(5):
        net.evaluate_codes(codes)
    for k in ("stem", "trunk", "head_conv", "head_fc", "encode"):
        ms, cnt = net.kernel_time_ms(k)
        out["class_" + k] = dict(total_ms=ms, launches=cnt, avg_us=1e3 * ms / max(cnt, 1))
        print("  %-10s %4d launches  avg %.1f us" % (k, cnt, 1e3 * ms / max(cnt, 1)))
    net.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
