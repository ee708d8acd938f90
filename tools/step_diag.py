#!/usr/bin/env python3
"""Per-step host timing of the self-play engine on the GPU box (1024 games, 2 x 512 leaves): step time
distribution and the apzh_advance / apzh_feed calls that take several times their median.  Run with
OMP_WAIT_POLICY=active to see the CPU-quota throttling spikes that the default (passive) removes."""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from alphapig_amd import weights
from alphapig_amd.policy_value_net import PolicyValueNet
from alphapig_amd.selfplay import SelfPlayEngine
prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
net = PolicyValueNet(15, 15, batch_size=512, n_blocks=10, n_filter=128, model_params=prm)
eng = SelfPlayEngine(net, 15, 15, 5, n_games=1024, n_playout=400, c_puct=5, temp=1.0, base_seed=0, pipeline=2)
import gc
eng.run_steps(30); gc.collect(); gc.freeze(); gc.disable()
# monkeypatch timers around pieces
import alphapig_amd.selfplay as sp
ts = []
orig_adv = eng._advance_group; orig_feed = eng.pool.feed
rec = {"adv": [], "feed": [], "wait": []}
def adv(ids):
    t = time.perf_counter(); r = orig_adv(ids); rec["adv"].append(time.perf_counter() - t); return r
def feed(ids, p, v):
    t = time.perf_counter(); r = orig_feed(ids, p, v); rec["feed"].append(time.perf_counter() - t); return r
eng._advance_group = adv; eng.pool.feed = feed
for i in range(200):
    t = time.perf_counter(); eng.run_steps(1); ts.append(time.perf_counter() - t)
ts = np.array(ts) * 1e3
print("step ms: median %.2f mean %.2f max %.2f" % (np.median(ts), ts.mean(), ts.max()))
print("slow steps (>1.5x median):", [(i, round(float(t), 1)) for i, t in enumerate(ts) if t > 1.5 * np.median(ts)][:30])
for k in ("adv", "feed"):
    a = np.array(rec[k]) * 1e3
    print(k, "calls", len(a), "median %.3f mean %.3f max %.3f" % (np.median(a), a.mean(), a.max()), "slow:", [(i, round(float(t), 1)) for i, t in enumerate(a) if t > 3 * np.median(a)][:20])
eng.close(); net.close()
