"""(Lives under tests/: it uses the CPU oracle, which only tests, smoke() and the cpu_baseline of bench.py may import.)
fp32 error of F(2x2,3x3) / F(4x4,3x3) Winograd trunk convolutions over the whole 20-layer residual
net, against the float64 oracle (CPU, numpy): the evidence behind trunk15_wino.h staying inside
the 1e-4 logit tolerance.  Uses oracle/ (test infrastructure) -- a tool, not part of the product path."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import net_ref
from alphapig_amd import weights

def wino_mats(m):
    if m == 2:
        Bt = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], float)
        G = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], float)
        At = np.array([[1,1,1,0],[0,1,-1,-1]], float)
    else:
        Bt = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], float)
        G = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], float)
        At = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], float)
    return Bt, G, At

def make_wino_conv(m):
    Bt, G, At = wino_mats(m)
    t = m + 2
    Bt32, At32 = Bt.astype(np.float32), At.astype(np.float32)
    def conv(x, w, b):
        n, ci, h, wd = x.shape
        co = w.shape[0]
        if w.shape[2] != 3 or x.dtype != np.float32 or ci != 128:
            return orig(x, w, b)
        U = np.einsum('ij,ocjk,lk->ocil', G, w.astype(np.float64), G).astype(np.float32)   # [co,ci,t,t]
        nt = -(-h // m)
        hp = nt * m + 2
        xp = np.zeros((n, ci, hp, hp), np.float32)
        xp[:, :, 1:1 + h, 1:1 + wd] = x
        out = np.zeros((n, co, nt * m, nt * m), np.float32)
        for ty in range(nt):
            for tx in range(nt):
                d = xp[:, :, ty*m:ty*m+t, tx*m:tx*m+t]
                # fp32 transforms (sequential adds, like a kernel would)
                V = np.einsum('ij,ncjk->ncik', Bt32, d).astype(np.float32)
                V = np.einsum('ncik,lk->ncil', V, Bt32).astype(np.float32)
                M = np.einsum('ncil,ocil->noil', V, U).astype(np.float32)   # fp32 accumulate (numpy pairwise-ish)
                Y = np.einsum('ij,nojk->noik', At32, M).astype(np.float32)
                Y = np.einsum('noik,lk->noil', Y, At32).astype(np.float32)
                out[:, :, ty*m:(ty+1)*m, tx*m:(tx+1)*m] = Y
        return out[:, :, :h, :wd] + b[None, :, None, None]
    return conv

orig = net_ref._conv
rs = np.random.RandomState(1)
for style in ('bench', 'reference'):
    prm = weights.init_params('resnet', 15, 15, 4, 10, 128, seed=0, style=style)
    # random-ish boards: 4 planes binary
    N = 8
    planes = np.zeros((N, 4, 15, 15), np.float32)
    for i in range(N):
        k = rs.randint(0, 120)
        cells = rs.permutation(225)[:k]
        for j, c in enumerate(cells):
            planes[i, j % 2, c // 15, c % 15] = 1
        if k: planes[i, 2, cells[-1] // 15, cells[-1] % 15] = 1
        planes[i, 3] = k % 2
    net_ref._conv = orig
    l64, p64, v64, t64 = net_ref.forward_resnet(prm, planes, 10, dtype=np.float64)[:4]
    l32, p32, v32, t32 = net_ref.forward_resnet(prm, planes, 10, dtype=np.float32)[:4]
    print(style, 'logit scale', np.abs(l64).max(), np.abs(v64).max())
    print(' direct fp32  : dlogit %.3g dv %.3g dp %.3g' % (np.abs(l32-l64).max(), np.abs(v32-v64).max(), np.abs(p32-p64).max()))
    for m in (2, 4):
        net_ref._conv = make_wino_conv(m)
        lw, pw, vw, tw = net_ref.forward_resnet(prm, planes, 10, dtype=np.float32)[:4]
        print(' winograd F(%d,3): dlogit %.3g dv %.3g dp %.3g' % (m, np.abs(lw-l64).max(), np.abs(vw-v64).max(), np.abs(pw-p64).max()))
