#!/usr/bin/env python3
"""CPU-only timing of the host tree (libalphapig_host.so): G games of 15x15 / n_playout 400 in lock step, one
`apzh_advance` + `apzh_feed` per step with priors of the shape a random-init net produces (near uniform + noise).
Prints thread-microseconds per playout -- what decides how many host cores a rank needs to keep its GPU busy
(one 1024-leaf step of the GPU takes ~4.3 ms).   usage: host_tree_bench.py [threads] [games] [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphapig_amd.treepool import TreePool, MOVE_READY

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 1
G = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1600
pool = TreePool(15, 15, 5, n_games=G, n_playout=400, c_puct=5, n_threads=threads)
rng = np.random.RandomState(0)
ids = np.arange(G, dtype=np.int32)
for g in range(G):
    pool.reset(g, 0)
probs_bank = rng.dirichlet(np.full(225, 20.0), size=(8, G)).astype(np.float32)
vals_bank = (rng.rand(8, G).astype(np.float32) - 0.5) * 0.2
t_adv = t_feed = 0.0
n_eval = 0
moves = 0
for s in range(steps):
    t0 = time.perf_counter()
    st, codes = pool.advance(ids)
    t1 = time.perf_counter()
    ready = np.nonzero(st == MOVE_READY)[0]
    if len(ready):                        # play the most visited move, keep the subtree (outside the timing)
        v, _ = pool.root_visits_dense(ids[ready])
        for g, row in zip(ready, v):
            ended, _, _ = pool.play_move(int(g), int(np.argmax(row)))
            if ended:
                pool.reset(int(g), 0)
            moves += 1
    if len(ready):
        t0b = time.perf_counter()
        st, codes = pool.advance(ids)
        t1 += time.perf_counter() - t0b
    need = ids[st == 1]
    t2 = time.perf_counter()
    pool.feed(need, probs_bank[s & 7][: len(need)], vals_bank[s & 7][: len(need)])
    t3 = time.perf_counter()
    t_adv += t1 - t0
    t_feed += t3 - t2
    n_eval += len(need)
print("threads %d games %d steps %d moves %d: advance %.3f ms/step, feed %.3f ms/step; %.2f thread-us per playout (advance %.2f + feed %.2f)"
      % (threads, G, steps, moves, t_adv / steps * 1e3, t_feed / steps * 1e3, (t_adv + t_feed) * threads / n_eval * 1e6,
         t_adv * threads / n_eval * 1e6, t_feed * threads / n_eval * 1e6))
