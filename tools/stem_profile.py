#!/usr/bin/env python3
"""The north_star target kernel on its target shape, for rocprofv3: the stem 3x3 convolution + folded BN + ReLU
(reference op: policy_value_net_mxnet.py:73) at 8192 x C_in x 15 x 15 -> 128 channels, C_in = 4 (the HBM-roofline
shape BASELINE.json names) and C_in = 9 (the real input).  Put the program directly after `--`:

  rocprofv3 --kernel-trace --stats --output-format csv -d OUT/trace -- python3 tools/stem_profile.py
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT/pmc_fetch -- python3 tools/stem_profile.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d OUT/pmc_write -- python3 tools/stem_profile.py

Prints the HIP-event timings of the same launches (apz_conv3x3_bench) as one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from alphapig_amd import weights  # noqa: E402
from alphapig_amd.policy_value_net import PolicyValueNet  # noqa: E402
from kernel_bench import synth  # noqa: E402


def main():
    n, out = 8192, {}
    for c_in in (4, 9):
        prm = weights.init_params("resnet", 15, 15, c_in, 1, 128, seed=0, style="bench")
        net = PolicyValueNet(15, 15, batch_size=n, n_blocks=1, n_filter=128, model_params=prm, c_in=c_in)
        _, planes = synth(n, 15, c_in)
        net.forward_planes(planes)
        ms = net.conv_bench(0, n, iters=30, warmup=5)
        alg = n * (c_in * 225 + 128 * 225) * 4 + 128 * c_in * 9 * 4
        out["c_in_%d" % c_in] = {"us_per_launch": ms * 1e3, "algorithmic_bytes": alg, "gb_per_s": alg / ms / 1e6,
                                 "frac_of_8TBs": alg / ms / 1e6 / 8000.0}
        net.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
