#!/usr/bin/env python3
"""Trunk-shape weight gradient through the Winograd domain, the two decompositions against each other and against
float64 (apz_wgrad_wino: by channel blocks, csrc/wgrad_wino2.h, the default; APZ_WGRAD_KERNEL=1: by position groups,
csrc/wgrad_wino.h).  Run twice on the GPU box: `wgrad_compare.py save FILE` under each setting, then
`wgrad_compare.py diff A B`."""
import sys
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIZES = (1, 7, 32, 33, 128, 515)


def save(path):
    import torch
    from alphapig_amd import hipconv
    out = {}
    for n in SIZES:
        g = torch.Generator(device="cuda").manual_seed(100 + n)
        x = torch.randn(n, 128, 15, 16, device="cuda", generator=g)
        dy = torch.randn(n, 128, 15, 16, device="cuda", generator=g)
        x[..., 15] = 0
        dy[..., 15] = 0
        L, hnd, stream = hipconv._ctx(x, hipconv.ROWS16)          # the Winograd-domain entry point at EVERY size
        dw = torch.empty(128, 128, 3, 3, device="cuda")              # (conv3x3_wgrad takes the direct kernel below 64 boards)
        hipconv._ck(L, L.apz_wgrad_wino(hnd, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, stream))
        torch.cuda.synchronize()
        out["dw%d" % n] = dw.cpu().numpy()
        if n <= 33:      # float64 reference: dw[co][ci][a][b] = sum x[n][ci][y + a - 1][x + b - 1] dy[n][co][y][x]
            xd = torch.nn.functional.pad(x[..., :15].double(), (1, 1, 1, 1))
            dyd = dy[..., :15].double()
            ref = torch.zeros(128, 128, 3, 3, dtype=torch.float64, device="cuda")
            for a in range(3):
                for b in range(3):
                    ref[:, :, a, b] = torch.einsum("nchw,nohw->oc", xd[:, :, a:a + 15, b:b + 15], dyd)
            out["ref%d" % n] = ref.cpu().numpy()
    np.savez(path, **out)
    print("saved", path)


def diff(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for n in SIZES:
        x, y = A["dw%d" % n], B["dw%d" % n]
        scale = np.abs(x).max()
        d = np.abs(x - y).max() / scale
        line = "n=%4d  max|A-B|/scale %.2e" % (n, d)
        if "ref%d" % n in A:
            r = A["ref%d" % n]
            ea, eb = np.abs(x - r).max() / scale, np.abs(y - r).max() / scale
            line += "   vs float64: A %.2e  B %.2e" % (ea, eb)
            bad += ea > 1e-5 or eb > 1e-5
        bad += d > 2e-5
        print(line)
    print("RESULT:", "MISMATCH" if bad else "ok")
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "save":
        save(sys.argv[2])
    else:
        sys.exit(diff(sys.argv[2], sys.argv[3]))
