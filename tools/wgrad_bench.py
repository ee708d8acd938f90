#!/usr/bin/env python3
"""Timing of the trunk-shape weight gradient (apz_wgrad_wino: Winograd domain; apz_conv3x3_wgrad: direct) on the GPU
box, HIP events over repeated launches, and of the other per-layer training kernels at the same batch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from alphapig_amd import hipconv
    out = {}
    for n in (128, 512):
        # ROT tensor pairs used in turn (ROT x 2 x 63 MB at 512 boards): more than the 256 MB Infinity Cache, so every
        # call reads its operands from HBM like a training step does -- with one pair the 512-board case measures a
        # cache-resident kernel (206 us instead of 250)
        ROT = 6
        xs, dys = [], []
        for _ in range(ROT):
            x = torch.randn(n, 128, 15, 16, device="cuda")
            x[..., 15] = 0
            dy = torch.randn(n, 128, 15, 16, device="cuda")
            dy[..., 15] = 0
            xs.append(x)
            dys.append(dy)
        w = torch.randn(128, 128, 3, 3, device="cuda") / 34
        turn = [0]

        def pair():
            turn[0] = (turn[0] + 1) % ROT
            return xs[turn[0]], dys[turn[0]]

        def timed(fn, iters=30):
            for _ in range(5):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e3 / iters

        r = {"wgrad": timed(lambda: hipconv.conv3x3_wgrad(*pair(), hipconv.ROWS16)),
             "wgrad_cache_resident": timed(lambda: hipconv.conv3x3_wgrad(xs[0], dys[0], hipconv.ROWS16)),
             "fwd": timed(lambda: hipconv.conv3x3_fwd(pair()[0], w, None, hipconv.ROWS16)),
             "bias_grad": timed(lambda: hipconv.bias_grad(pair()[1], hipconv.ROWS16))}
        mfma_flops = n * 9216 * 2048.0
        r["wgrad_executed_tflops"] = mfma_flops / r["wgrad"] / 1e6
        r["wgrad_frac_of_157"] = r["wgrad_executed_tflops"] / 157.3
        out["n%d" % n] = r
        print(n, {k: round(v, 3) for k, v in r.items()}, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
