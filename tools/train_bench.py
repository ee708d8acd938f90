#!/usr/bin/env python3
"""Training-step timing on the GPU box: interim trainer with torch (MIOpen) convolutions vs the
hand-written HIP convolution primitives (forward, dgrad, wgrad), 10-block net, 15x15."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphapig_amd import weights  # noqa: E402
from alphapig_amd.train import TorchTrainer  # noqa: E402


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="128,512")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    rs = np.random.RandomState(0)
    prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
    out = {}
    for B in [int(x) for x in args.batches.split(",")]:
        states = (rs.rand(B, 9, 15, 15) > 0.7).astype(np.float32)
        pis = rs.dirichlet(np.ones(225), size=B).astype(np.float32)
        zs = rs.choice([-1.0, 1.0], size=B).astype(np.float32)
        for backend in ("torch", "hip-direct", "hip-wino", "hip16", "hip"):
            # "hip": the default choice (from 192 boards: the whole trunk on HIP kernels in the padded-row layout =
            # "hip16"); "hip-direct" / "hip-wino": dense tensors, torch BatchNorm, direct / Winograd forward + dgrad
            os.environ["APZ_TRAIN_CONV"] = {"hip-direct": "direct", "hip-wino": "wino"}.get(backend, "auto")
            tr = TorchTrainer(prm, "resnet", n_blocks=10, batch_size=B, device="cuda",
                              conv_backend="torch" if backend == "torch" else "hip",
                              trunk_backend={"hip16": "hip16", "hip": None}.get(backend, "torch"))
            for _ in range(3):
                tr.train_step(states, pis, zs, 1e-3)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(args.steps):
                tr.train_step(states, pis, zs, 1e-3)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t) / args.steps
            flops = 3 * 2.0 * B * (9 * 128 + 20 * 128 * 128) * 9 * 225       # fwd + dgrad + wgrad of the 3x3 convs
            out["B%d_%s" % (B, backend)] = {"ms_per_step": ms, "conv_tflops_equiv": flops / ms / 1e9}
            print("batch %4d  convs=%-10s  %.2f ms/step   (3x3-conv work %.1f TFLOP/s equivalent)" %
                  (B, backend, ms, flops / ms / 1e9), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
