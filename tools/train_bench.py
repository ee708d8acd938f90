#!/usr/bin/env python3
"""Training-step timing on the GPU box: the product trainer (alphapig_amd/train.py, every operator a HIP kernel of
this repository) beside the same graph in PyTorch (MIOpen convolutions; tests/torch_trainer.py, the comparator the
trainer is tested against), 10-block net, 15x15."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from alphapig_amd import weights  # noqa: E402
from alphapig_amd.train import HipTrainer  # noqa: E402


def main():
    import torch
    from torch_trainer import TorchTrainer
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="128,512")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-torch", action="store_true")
    args = ap.parse_args()
    rs = np.random.RandomState(0)
    prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
    out = {}
    for B in [int(x) for x in args.batches.split(",")]:
        states = (rs.rand(B, 9, 15, 15) > 0.7).astype(np.float32)
        pis = rs.dirichlet(np.ones(225), size=B).astype(np.float32)
        zs = rs.choice([-1.0, 1.0], size=B).astype(np.float32)
        for backend in ("hip",) if args.no_torch else ("torch", "hip"):
            if backend == "hip":
                tr = HipTrainer(prm, "resnet", n_blocks=10, batch_size=B)
            else:
                tr = TorchTrainer(prm, "resnet", n_blocks=10, batch_size=B, device="cuda")
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
            print("batch %4d  %-6s  %.2f ms/step   (3x3-conv work %.1f TFLOP/s equivalent)" %
                  (B, backend, ms, flops / ms / 1e9), flush=True)
            if backend == "hip":     # what epochs 2 .. n of a policy_update pay: the mini-batch is on the device already
                batch = tr.upload(states, pis, zs)
                tr.train_step(batch, None, None, 1e-3)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(args.steps):
                    tr.train_step(batch, None, None, 1e-3)
                torch.cuda.synchronize()
                ms = 1e3 * (time.perf_counter() - t) / args.steps
                out["B%d_hip_resident_batch" % B] = {"ms_per_step": ms}
                print("batch %4d  hip     %.2f ms/step on an uploaded mini-batch (HipTrainer.upload: policy_update's epochs share one)" %
                      (B, ms), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
