"""SelfPlayEngine: G concurrent self-play games, ONE coalesced leaf batch per step.

Each game slot runs exactly the reference episode (Game_AI.start_self_play, reference
game_ai.py:70-139: 9 % forced two-ply opening, get_action(temp, return_prob=1) per ply,
z assignment) with the reference search (mcts_alphaZero.py:141-218: n_playout sequential
playouts, visit-count softmax, Dirichlet(0.3) mixed into the sampling distribution, subtree
reuse).  The reference evaluates one leaf per forward; here every game keeps at
most one leaf in flight, so the G pending leaves of a step form one contiguous batch for the
HIP evaluator while every game's tree stays bit-identical to a sequential run.

Game k (global index) draws from np.random.RandomState(base_seed + k) and
random.Random(base_seed + k): the same streams as `np.random.seed` / `random.seed` in a
sequential reference run of that game.

Host/GPU overlap: the slots are split into `pipeline` groups; while the evaluator works on one
group's leaves (in a worker thread -- ctypes releases the GIL) the native tree pool advances the
next group.
"""
import collections
import random as _random
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .game_ai import draw_forced_opening, one_hot_pi
from .rootsample import LegacyRngBank, sample_moves
from .treepool import TreePool, NEED_EVAL, MOVE_READY

Episode = collections.namedtuple("Episode", "index moves movers codes pis zs winner")


class PlanesEvaluator(object):
    """Adapter: wraps `fn(planes[n,9,H,W] float32) -> (probs[n,HW], values[n])` as an evaluator
    with evaluate_codes().  Used for CPU tests of the scheduler with a stand-in network."""

    def __init__(self, fn, pool):
        self._fn, self._pool = fn, pool

    def evaluate_codes(self, codes):
        p, v = self._fn(self._pool.codes_to_planes(codes, 9))
        return np.ascontiguousarray(p, dtype=np.float32), np.ascontiguousarray(v, dtype=np.float32).reshape(-1)


class _Slot(object):
    __slots__ = ("index", "rng", "pyrnd", "codes", "pis", "movers", "active")

    def __init__(self):
        self.active = False


class SelfPlayEngine(object):
    def __init__(self, evaluator, board_width=15, board_height=15, n_in_row=5, n_games=1024, n_playout=400,
                 c_puct=5, temp=1.0, base_seed=0, n_threads=0, pipeline=2, noise_alpha=0.3, noise_eps=0.25,
                 forced_opening=True, index_offset=0, index_stride=1, sampler="host", temp_schedule=None):
        self.pool = TreePool(board_width, board_height, n_in_row, n_games=n_games, n_playout=n_playout,
                             c_puct=c_puct, prior_is_f32=True, n_threads=n_threads)
        self.evaluator = evaluator if hasattr(evaluator, "evaluate_codes") else PlanesEvaluator(evaluator, self.pool)
        # An evaluator with stream-ordered slots (PolicyValueNet.evaluate_codes_slot) lets every
        # pipeline group keep one batch queued on the SAME HIP stream: group B's kernels start the
        # moment group A's last kernel ends, while kernels never run concurrently (clean timings).
        self._slotted = hasattr(self.evaluator, "evaluate_codes_slot")
        # ... and one whose submit returns at once (submit_codes_slot / wait_slot) needs no worker threads at all: the
        # scheduler thread queues group A's batch, advances group B while the GPU works, and only then waits for A --
        # no GIL hand-offs between threads inside a step (they cost more than the whole host side of a 32-leaf step)
        self._async = self._slotted and hasattr(self.evaluator, "submit_codes_slot") and hasattr(self.evaluator, "wait_slot")
        self.G, self.hw = int(n_games), board_width * board_height
        self.n_playout, self.temp = int(n_playout), temp
        # The reference plays every ply at one constant `temp`.  Optional extension
        # (BASELINE config 5): temp_schedule = [(first_ply, temp), ...] ascending, or callable(ply).
        self.temp_schedule = temp_schedule
        self.base_seed = int(base_seed)
        self.noise_alpha, self.noise_eps = noise_alpha, noise_eps
        if forced_opening and (board_width != 15 or board_height != 15):
            # the reference's opening book is written for the 15-wide board (game_ai.py:77-78);
            # on any other board its moves are illegal and the reference raises at list.remove
            raise ValueError("forced_opening needs a 15x15 board; pass forced_opening=False")
        self.forced_opening = forced_opening
        self.pipeline = max(1, int(pipeline))
        # multi-GPU sharding: this engine owns global games offset, offset+stride, ... (dist.py)
        self.index_offset, self.index_stride = int(index_offset), int(index_stride)
        # "host" (default): the reference's NumPy legacy stream per game, bit for bit, drawn for all ready games in one
        #         native call (rootsample.py: apzh_root_sample);
        # "numpy": the same draws through np.random.RandomState itself, one game at a time (the reference's own calls;
        #         what "host" is tested against);
        # "gpu":  root_sample_kernel (statistical parity only)
        if sampler not in ("host", "numpy", "gpu"):
            raise ValueError("sampler must be 'host', 'numpy' or 'gpu'")
        self.sampler = sampler
        self.rng_bank = LegacyRngBank(self.G) if sampler == "host" else None
        self._sample_threads = max(1, min(8, int(n_threads) if n_threads else 4))
        self._sample_step = 0
        self.slots = [_Slot() for _ in range(self.G)]
        self.slot_games = np.zeros(self.G, dtype=np.int64)      # games finished per slot (steady-state detection)
        self.next_index = 0
        self.finished = []
        self.stats = collections.Counter()
        self.timers = collections.Counter()
        self.step_times = None          # bench.py sets a list: wall time of every scheduler round lands there
        # optional observer callable(slot_ids, codes, probs, values), called once per evaluated batch right before the
        # outputs are fed to the trees: which slot asked for which position and what the evaluator answered (the parity
        # tests record the games they replay through the sequential oracle with it; None costs nothing)
        self.tap = None
        self._limit = None
        # optional threading.Event: while it is CLEAR run_steps starts no new scheduler round (evaluations already in flight
        # finish; the next round begins when it is set again) -- the training pipeline's trainer thread holds it clear
        # during a policy update on the same GPU (pipeline.TrainPipeline, exclusive_updates)
        self.gate = None
        self.timers["gate_s"] = 0.0
        self._active_epoch = 0          # bumped whenever a slot's `active` flag changes
        self._grp_cache = {}            # group (tuple of slot ids) -> (epoch, its active slots as an int32 array)
        n_workers = min(self.pipeline, getattr(self.evaluator, "n_slots", 1)) if self._slotted else 1
        # one worker thread per slot; a worker serves its own groups in order
        self._n_slots = max(1, n_workers)
        if self.pipeline > self._n_slots:           # two groups would share a slot: keep the one-worker-per-slot queues
            self._async = False
        self._exec = ([ThreadPoolExecutor(max_workers=1) for _ in range(n_workers)]
                      if self.pipeline > 1 and not self._async else None)

    # ---- slot life cycle -------------------------------------------------------------------
    def _start_game(self, s):
        slot = self.slots[s]
        if self._limit is not None and self.next_index >= self._limit:
            if slot.active:
                self._active_epoch += 1
            slot.active = False
            return False
        k = self.index_offset + self.next_index * self.index_stride     # global game index
        self.next_index += 1
        slot.index = k
        self.set_slot_rng(s, self.base_seed + k)
        slot.pyrnd = _random.Random(self.base_seed + k)
        slot.codes, slot.pis, slot.movers = [], [], []
        if not slot.active:
            self._active_epoch += 1
        slot.active = True
        self.pool.reset(s, 0)
        forced = draw_forced_opening(slot.pyrnd) if self.forced_opening else None
        if forced is not None:
            for mv in forced:
                self._record(s, one_hot_pi(self.hw, mv))
                self.pool.play_move(s, mv)
            self.stats["forced_openings"] += 1
        return True

    def set_slot_rng(self, s, seed_or_state):
        """Slot s draws its root samples from np.random.RandomState(seed) -- or from a copy of the given RandomState's
        current state (tests replay the reference's recorded streams that way)."""
        slot = self.slots[s]
        if self.rng_bank is not None:
            slot.rng = None
            if hasattr(seed_or_state, "get_state"):
                self.rng_bank.set_state(s, seed_or_state)
            elif 0 <= int(seed_or_state) < 2 ** 32:
                self.rng_bank.seed(s, int(seed_or_state))
            else:
                self.rng_bank.set_state(s, np.random.RandomState(seed_or_state))
        else:
            slot.rng = seed_or_state if hasattr(seed_or_state, "get_state") else np.random.RandomState(seed_or_state)

    def _record(self, s, pi):
        slot = self.slots[s]
        slot.codes.append(self.pool.codes(s))
        slot.pis.append(pi)
        slot.movers.append(self.pool.status(s)[0])

    def _finish_game(self, s, winner):
        slot = self.slots[s]
        movers = np.array(slot.movers)
        z = np.zeros(len(movers))
        if winner != -1:
            z[movers == winner] = 1.0
            z[movers != winner] = -1.0
        moves, _ = self.pool.history(s)
        self.finished.append(Episode(slot.index, moves.astype(np.int32), movers.astype(np.int8),
                                     np.stack(slot.codes), np.stack(slot.pis), z, winner))
        self.stats["games"] += 1
        self.stats["plies"] += len(movers)
        self.slot_games[s] += 1

    def _temp_for(self, ply):
        sch = self.temp_schedule
        if sch is None:
            return self.temp
        if callable(sch):
            return float(sch(ply))
        t = self.temp
        for first, val in sch:
            if ply >= first:
                t = val
        return t

    # ---- one move of every MOVE_READY slot (mcts_alphaZero.py:151-157, :187-203) ------------
    def _play_ready(self, ready):
        visits, _ = self.pool.root_visits_dense(ready)
        if self.sampler == "gpu":
            self._sample_step += 1
            temps = [self._temp_for(len(self.slots[int(s)].movers)) for s in ready]
            pis = np.empty((len(ready), self.hw), dtype=np.float32)
            moves = np.empty(len(ready), dtype=np.int32)
            # one key per row: (global game index, ply).  A game's noise then depends on the seed, the game and the
            # ply only -- not on the batch row, the step counter or the rank (ranks share base_seed and step counters)
            keys = np.array([(self.slots[int(s)].index << 20) | len(self.slots[int(s)].movers) for s in ready], dtype=np.uint64)
            for t in sorted(set(temps)):            # one launch per distinct temperature
                sel = np.array([i for i, ti in enumerate(temps) if ti == t])
                pis[sel], moves[sel] = self.evaluator.sample_moves(
                    visits[sel], temp=t, alpha=self.noise_alpha, eps=self.noise_eps,
                    seed=self.base_seed, step=self._sample_step, keys=keys[sel])
            for pi, move, s in zip(pis, moves, ready):
                s = int(s)
                self._record(s, pi.astype(np.float64))
                ended, winner, _ = self.pool.play_move(s, int(move))
                self.stats["moves"] += 1
                if ended:
                    self._finish_game(s, winner)
                    self._start_game(s)
            return
        if self.sampler == "host":
            temps = np.array([self._temp_for(len(self.slots[int(s)].movers)) for s in ready], dtype=np.float64)
            pis, moves = sample_moves(self.rng_bank, ready, visits, temps, alpha=self.noise_alpha, eps=self.noise_eps,
                                      with_noise=True, want_pi=True, n_threads=self._sample_threads)
            # record (state, pi, player) and play the move of every ready game in one native call
            codes_b, movers_b, ended_b, winner_b = self.pool.play_moves(ready, moves)
            self.stats["moves"] += len(ready)
            for i, s in enumerate(ready):
                slot = self.slots[int(s)]
                slot.codes.append(codes_b[i])
                slot.pis.append(pis[i])
                slot.movers.append(int(movers_b[i]))
                if ended_b[i]:
                    self._finish_game(int(s), int(winner_b[i]))
                    self._start_game(int(s))
            return
        for row, s in zip(visits, ready):
            s = int(s)
            slot = self.slots[s]
            acts = np.flatnonzero(row >= 0)
            x = 1.0 / self._temp_for(len(slot.movers)) * np.log(row[acts].astype(np.int64) + 1e-10)
            probs = np.exp(x - np.max(x))
            probs /= np.sum(probs)
            pi = np.zeros(self.hw)
            pi[acts] = probs
            noise = slot.rng.dirichlet(self.noise_alpha * np.ones(len(probs)))
            move = int(slot.rng.choice(acts, p=(1.0 - self.noise_eps) * probs + self.noise_eps * noise))
            self._record(s, pi)
            ended, winner, _ = self.pool.play_move(s, move)
            self.stats["moves"] += 1
            if ended:
                self._finish_game(s, winner)
                self._start_game(s)

    # ---- scheduling ------------------------------------------------------------------------
    def _advance_group(self, ids, fed=None):
        """Advance the slots in `ids` until each one waits for an evaluation (or went idle).
        fed = (ids, probs, values) of the group's finished evaluation, not fed to the trees yet: when those are exactly
        the group's active slots (the steady state) feed + first advance are ONE native call (apzh_feed_advance).
        -> (eval_ids, eval_codes)"""
        t0 = time.perf_counter()
        # the group's active slots: the same array object as long as no slot changed its `active` flag (the steady state of a
        # long run) -- a round of BASELINE config 2 is ~125 us, of which rebuilding and comparing these 32-entry arrays, the
        # masks and the concatenations below were ~15 (profiles/r06_config2.md)
        key = ids if isinstance(ids, tuple) else tuple(ids)
        hit = self._grp_cache.get(key)
        if hit is None or hit[0] != self._active_epoch:
            hit = (self._active_epoch, np.array([s for s in key if self.slots[s].active], dtype=np.int32))
            self._grp_cache[key] = hit
        ids = hit[1]
        fused = fed is not None and (fed[0] is ids or (len(fed[0]) == len(ids) and
                                                       np.array_equal(np.asarray(fed[0], dtype=np.int32), ids)))
        if fed is not None and not fused:
            self.pool.feed(fed[0], fed[1], fed[2])
        eval_ids, eval_codes = [], []
        while len(ids):
            if fused:
                st, codes = self.pool.feed_advance(ids, fed[1], fed[2])
                fused = False
            else:
                st, codes = self.pool.advance(ids)
            if not eval_ids and int(st.min()) == NEED_EVAL == int(st.max()):     # every slot waits for its leaf: nothing to sort out
                self.timers["host_s"] += time.perf_counter() - t0
                return ids, codes
            need = st == NEED_EVAL
            if need.any():
                eval_ids.append(ids[need])
                eval_codes.append(codes[need])
            ready = ids[st == MOVE_READY]
            if len(ready) == 0:
                break
            t1 = time.perf_counter()
            self._play_ready(ready)
            self.timers["moves_s"] += time.perf_counter() - t1
            ids = np.array([s for s in ready if self.slots[s].active], dtype=np.int32)
        self.timers["host_s"] += time.perf_counter() - t0
        if not eval_ids:
            return np.zeros(0, np.int32), np.zeros((0, self.pool.code_stride), np.uint8)
        return np.concatenate(eval_ids), np.concatenate(eval_codes)

    def _evaluate(self, codes, slot=0):
        t0 = time.perf_counter()
        if self._slotted:
            out = self.evaluator.evaluate_codes_slot(slot, codes)
        else:
            out = self.evaluator.evaluate_codes(codes)
        self.timers["eval_s"] += time.perf_counter() - t0
        return out

    def _dispatch(self, gi, ids, codes):
        """Start the evaluation of one group's leaves -> a ticket for _collect."""
        if self._async and len(ids) <= getattr(self.evaluator, "batchsize", len(ids)):
            slot = gi % self._n_slots
            t0 = time.perf_counter()
            n = self.evaluator.submit_codes_slot(slot, codes)
            self.timers["eval_s"] += time.perf_counter() - t0
            return ("slot", slot, n, ids, codes)
        if self._exec is not None:
            w = gi % len(self._exec)
            return ("future", self._exec[w].submit(self._evaluate, codes, w), ids, codes)
        return ("done", self._evaluate(codes), ids, codes)

    def _collect(self, ticket):
        """-> (probs, values, ids) of a dispatched group (blocks until the GPU is done with it)."""
        if ticket[0] == "slot":
            t0 = time.perf_counter()
            p, v = self.evaluator.wait_slot(ticket[1], ticket[2])
            self.timers["eval_s"] += time.perf_counter() - t0
            ids = ticket[3]
        elif ticket[0] == "future":
            p, v = ticket[1].result()
            ids = ticket[2]
        else:
            (p, v), ids = ticket[1], ticket[2]
        if self.tap is not None:
            self.tap(ids, ticket[-1], p, v)
        return p, v, ids

    def _groups(self):
        return [tuple(range(g, self.G, self.pipeline)) for g in range(self.pipeline)]

    def run_steps(self, n_steps, total_games=None):
        """Run n_steps scheduler rounds; in one round every active slot gets exactly one leaf
        evaluated (terminal-leaf playouts and move selection ride along on the host).  Slots are
        (re)seeded with fresh games until `total_games` have been started (None = unlimited).
        -> number of leaf evaluations done."""
        self._limit = total_games
        for s in range(self.G):
            if not self.slots[s].active:
                self._start_game(s)
        leafs = 0
        groups = self._groups()
        inflight = {}
        step_times = getattr(self, "step_times", None)      # measurement hook (bench.py): wall time of every scheduler round
        gate = self.gate
        for _ in range(n_steps):
            busy = False
            if gate is not None and not gate.is_set():
                t_gate = time.perf_counter()
                gate.wait()
                self.timers["gate_s"] += time.perf_counter() - t_gate
            t_step = time.perf_counter()
            for gi, grp in enumerate(groups):
                fed = None
                if gi in inflight:                       # finish this group's previous evaluation
                    p, v, ids = self._collect(inflight.pop(gi))
                    fed = (ids, p, v)
                    leafs += len(ids)
                ids, codes = self._advance_group(grp, fed)   # (feeds first) overlaps the other groups' evaluations
                if len(ids):
                    busy = True
                    inflight[gi] = self._dispatch(gi, ids, codes)
            if step_times is not None:
                step_times.append(time.perf_counter() - t_step)
            if not busy:
                break
        for gi in sorted(inflight):
            p, v, ids = self._collect(inflight[gi])
            self.pool.feed(ids, p, v)
            leafs += len(ids)
        self.stats["leaf_evals"] += leafs
        self.stats["rounds"] += n_steps
        return leafs

    def play_games(self, total_games, max_steps=None, progress=None):
        """Play `total_games` complete episodes -> list[Episode] ordered by game index.
        progress: optional callable(engine) invoked about every 20 s (long GPU runs must keep
        printing)."""
        steps = 0
        last = time.perf_counter()
        while self.stats["games"] < total_games:
            n = self.run_steps(64, total_games)
            steps += 64
            if progress is not None and time.perf_counter() - last > 20.0:
                progress(self)
                last = time.perf_counter()
            if n == 0 and not any(s.active for s in self.slots):
                break
            if max_steps is not None and steps >= max_steps:
                break
        out = sorted(self.finished, key=lambda e: e.index)
        return out

    def terminal_playouts(self):
        return sum(self.pool.stats(s)["terminal_playouts"] for s in range(self.G))

    def close(self):
        if self._exec is not None:
            for ex in self._exec:
                ex.shutdown(wait=True)
            self._exec = None
        self.pool.close()


def episodes_to_tuples(episodes, pool):
    """Flatten episodes to training tuples (state planes float32 [T,9,H,W], pi [T,HW], z [T])
    -- the (s, pi, z) triples of game_ai.py:139."""
    codes = np.concatenate([e.codes for e in episodes])
    pis = np.concatenate([e.pis for e in episodes])
    zs = np.concatenate([e.zs for e in episodes])
    return pool.codes_to_planes(codes, 9), pis, zs
