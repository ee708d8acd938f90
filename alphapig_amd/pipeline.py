"""TrainPipeline: the AlphaZero loop around the batched engine -- the consumer of the hot path
(reference train_mxnet.py:38-300; SURVEY.md lists it as the "next" rows f1/f3/f4 wired together).

Same configuration keys as the reference's conf/train_config.yaml (board_width, board_height,
n_in_row, learn_rate, lr_multiplier, temp, n_playout, c_puct, buffer_size, batch_size, epochs,
kl_targ, check_freq, pure_mcts_playout_num, game_batch_num, play_batch_size, sgf_dir) plus
    concurrent_games   self-play games in flight on the GPU (default 1024)
    sgf_batches        how many leading batches replay SGF records instead of searching (the
                       reference hard-codes 4000, train_mxnet.py:270; default 0 = none)
    n_blocks, n_filter network size (the reference hard-codes 10 / 128, train_mxnet.py:79-91)

One `game batch` of the reference = ONE self-play game followed by a policy update; here a batch
collects `play_batch_size` games from the engine, which keeps `concurrent_games` running so that
finished games are drawn from a continuously full GPU batch.  After every update the evaluator
is re-folded with the new weights and the engine's slots keep playing with them (their trees
were built under the previous weights -- like the reference's, whose tree is also reused across
the update that happens between two games only at game boundaries; here the switch can fall
inside a game).

Multi-GPU (BASELINE configs[4]: "competition-strength self-play for the train_mxnet.py pipeline" on 8 GPUs):
one process per GPU, `distributed=True` (default: on when alphapig_amd.dist holds a process group).  Per game batch
    every rank   self-plays its shard of the global game sequence (rank r owns games r, r + R, ...) until
                 `play_batch_size` of ITS games have finished, then joins ONE `dist.all_gather_tuples` of [codes | pi | z];
    rank 0       expands the gathered tuples (rank-major order), augments them (get_equi_data) into THE replay buffer,
                 runs `policy_update` (KL monitor and lr_multiplier live here only), saves / evaluates on schedule;
    every rank   receives rank 0's verdict (`dist.broadcast_floats`: updated?, loss, entropy, kl, lr_multiplier) and, after
                 an update, the new weights (`dist.broadcast_params`: one flat 12.7 MB RCCL broadcast, the cross-rank form of
                 the reference's predict-module re-sync, policy_value_net_mxnet.py:295-297) -> `load_device_params`
                 (device to device) and goes on playing with them.
The reference has no multi-process collector (train_mxnet.py:182-191 is commented out); the loop is its `run`
(train_mxnet.py:265-300) with collect -> all-gather -> update -> broadcast.

Two schedules (`async_update` in the configuration; default True):
  lock step    (False) the loop above, literally: every rank waits while rank 0 trains / evaluates (round 3; measured on
               one MI355X: the update is 55 % of the wall clock and self-play falls to 39 % of its rate; on N ranks every
               GPU idles through every update).
  asynchronous (True)  nobody waits for the trainer.  Self-play runs in ROUNDS of `round_seconds` wall clock (or
               `round_steps` engine steps); at the end of a round every rank hands over the games it finished since the
               last one (any number, also none) in one all-gather and receives a 9-float header from rank 0; weights
               travel only in rounds in which rank 0 has a NEW version.  On rank 0 the trainer lives in its own thread
               with its own HIP stream, its own evaluator handle (KL monitor, arena) and its own HipTrainer: it takes
               the gathered game batches from a queue, puts them into the replay buffer, runs `policy_update` -- one per
               game batch like the reference, never more; batches that arrive while an update runs are merged into the
               next one and counted as `updates_skipped` -- saves / evaluates on the reference's schedule and publishes a
               snapshot of the weights, which the next round broadcasts.  The SGF bootstrap batches (`sgf_batches`,
               train_mxnet.py:270-271) run on the trainer thread first, one record + one update per batch, while the ranks
               already self-play; they count as game batches.  Staleness bound: a rank plays with weights that
               are at most (one round + one policy_update + one broadcast) older than the trainer's; the switch can fall
               inside a game, as in lock step.  `max_update_share` < 1 makes the trainer pause after an update so that it
               is busy at most that share of the time (on ONE GPU self-play and training share the device: the
               reference's one-update-per-game rule costs 0.36 - 0.55 of it; see profiles/r04_train_loop_15x15.log).
"""
import logging
import os
import queue
import random
import threading
import time

import numpy as np

from . import dist, sgf
from .arena import Arena, win_ratio
from .augment import get_equi_data
from .game import Board, Game
from .policy_value_net import PolicyValueNet
from .selfplay import SelfPlayEngine, episodes_to_tuples
from .train import policy_update

_logger = logging.getLogger(__name__)


class ReplayBuffer(object):
    """The reference's `deque(maxlen=buffer_size)` of (state, pi, z) tuples (train_mxnet.py:59) with O(1) random access:
    `random.sample` on the deque costs O(buffer) per mini-batch (the reference's buffer holds 2 198 800 entries), a list
    used as a ring does not.  Index 0 is the oldest entry, as in the deque."""

    def __init__(self, maxlen):
        self.maxlen = int(maxlen)
        self._items = []
        self._head = 0                   # position of the oldest entry once the ring is full

    def __len__(self):
        return len(self._items)

    def __getitem__(self, i):
        n = len(self._items)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return self._items[(self._head + i) % n] if n == self.maxlen else self._items[i]

    def append(self, x):
        if len(self._items) < self.maxlen:
            self._items.append(x)
        else:
            self._items[self._head] = x
            self._head = (self._head + 1) % self.maxlen

    def extend(self, xs):
        for x in xs:
            self.append(x)

    def __iter__(self):
        for i in range(len(self._items)):
            yield self[i]

    def sample(self, rng, k):
        """`rng.sample(list(buffer), k)` -- the same draws from the same generator state -- without the copy."""
        return [self[j] for j in rng.sample(range(len(self._items)), k)]


class TrainPipeline(object):
    def __init__(self, conf, init_model=None, policy_value_net=None, device=0, seed=0, distributed=None, trainer=None,
                 eval_net=None):
        """policy_value_net: an evaluator to use instead of building the HIP PolicyValueNet (needs evaluate_codes, params /
        set_params; policy_value for the KL monitor on rank 0).  trainer: an object with HipTrainer's interface
        (train_step, get_params; optionally sync_evaluator) to use instead of creating a HipTrainer -- the CPU tests pass
        stand-ins for both.  eval_net (asynchronous schedule, rank 0): the trainer thread's own evaluator for the KL
        monitor, checkpoints and the arena; default: a second HIP PolicyValueNet, or `policy_value_net` itself when one
        was passed in.  distributed: None = on iff alphapig_amd.dist holds a process group."""
        self.async_update = bool(conf.get("async_update", True))
        self.round_seconds = float(conf.get("round_seconds", 0.5))
        self.round_steps = conf.get("round_steps")
        self.max_update_share = float(conf.get("max_update_share", 1.0))
        # exclusive_updates (default off): rank 0's self-play loop starts no new round while its trainer thread is inside a
        # policy_update (the in-flight forwards finish).  Measured on one MI355X with the reference's configuration
        # (profiles/r06_train_loop_exclusive.log / _interleaved.log): an update takes 45 ms instead of 130 (its ~200 small
        # launches per epoch no longer queue behind the self-play forwards' 60-us full-chip kernels), but self-play is held
        # 12 % of the wall clock and ends at 277 k leaf evaluations/s against 305 k interleaved: interleaving wins on
        # throughput; the switch is for jobs that want the shortest update latency.
        self.exclusive_updates = bool(conf.get("exclusive_updates", False))
        self._gpu_gate = threading.Event()
        self._gpu_gate.set()
        self._custom_net = policy_value_net is not None
        self._eval_net = eval_net
        self._device = device
        self.distributed = dist.is_active() if distributed is None else bool(distributed)
        self.rank, self.world = dist.rank_world() if self.distributed else (0, 1)
        self._trainer_override = trainer
        self._seed = seed
        self._train_steps = 0
        self.board_width, self.board_height = conf["board_width"], conf["board_height"]
        self.n_in_row = conf["n_in_row"]
        self.learn_rate, self.lr_multiplier = conf["learn_rate"], conf.get("lr_multiplier", 1.0)
        self.temp, self.n_playout, self.c_puct = conf["temp"], conf["n_playout"], conf["c_puct"]
        self.batch_size = conf["batch_size"]
        self.data_buffer = ReplayBuffer(conf["buffer_size"])
        self.play_batch_size = conf.get("play_batch_size", 1)
        self.epochs, self.kl_targ = conf["epochs"], conf["kl_targ"]
        self.check_freq, self.game_batch_num = conf["check_freq"], conf["game_batch_num"]
        self.pure_mcts_playout_num = conf["pure_mcts_playout_num"]
        self.eval_games = conf.get("eval_games", 10)
        self.best_win_ratio = 0.0
        self.sgf_batches = conf.get("sgf_batches", 0)
        self._sgf_home = conf.get("sgf_dir")
        self._training_data = []
        if self.sgf_batches and self._sgf_home and os.path.isdir(self._sgf_home):
            self._training_data = sgf.get_files_as_list(self._sgf_home)
            random.shuffle(self._training_data)
        self.model_dir = conf.get("model_dir", "./logs")
        self.board = Board(width=self.board_width, height=self.board_height, n_in_row=self.n_in_row)
        self.game = Game(self.board)
        concurrent = conf.get("concurrent_games", 1024)
        self.policy_value_net = policy_value_net or PolicyValueNet(
            self.board_width, self.board_height, max(self.batch_size, (concurrent + 1) // 2),
            n_blocks=conf.get("n_blocks", 10), n_filter=conf.get("n_filter", 128), model_params=init_model,
            device=device, seed=seed)
        self.engine = SelfPlayEngine(self.policy_value_net, self.board_width, self.board_height, self.n_in_row,
                                     n_games=concurrent, n_playout=self.n_playout, c_puct=self.c_puct,
                                     temp=self.temp, base_seed=seed, pipeline=2,
                                     forced_opening=(self.board_width == 15 and self.board_height == 15),
                                     index_offset=self.rank, index_stride=self.world)
        self.keep_replica_buffers = bool(conf.get("replica_buffers", False))    # every rank keeps the replay buffer (host RAM x world)
        self._taken = 0
        self._rng = random.Random(seed)
        self.episode_len = 0
        self.history = []
        # asynchronous schedule
        self.weights_version = 0
        self.weight_broadcasts = 0
        self.update_intervals = []              # rank 0: (start, end) wall clock of every policy_update (time.time())
        self.round_log = []                     # every rank: (time.time(), leaf evaluations so far) at the end of each round
        self.updates_done = self.updates_skipped = 0
        self._train_q = None
        self._fresh = None
        self._lock = threading.Lock()
        self._trainer_error = None
        self._sgf_done = threading.Event()     # asynchronous schedule: set by rank 0's trainer thread behind the SGF bootstrap
        self._last = (0.0, 0.0, 0.0)

    # ---- data collection -----------------------------------------------------------------
    def collect_selfplay_data_ai(self, n_games=1):
        """train_mxnet.py:170-180, batched: take the next n finished games from the engine (of THIS rank's shard; with
        several ranks the batch is the rank-major concatenation of every rank's n games)."""
        while len(self.engine.finished) < n_games:           # completion order, like a game queue
            self.engine.run_steps(16)
        eps = self.engine.finished[:n_games]
        del self.engine.finished[:n_games]
        self._taken += n_games
        self.episode_len = int(np.mean([len(e.moves) for e in eps]))
        if not self.distributed:
            states, pis, zs = episodes_to_tuples(eps, self.engine.pool)
        else:
            codes = np.concatenate([e.codes for e in eps])
            pis = np.concatenate([e.pis for e in eps]).astype(np.float32)
            zs = np.concatenate([e.zs for e in eps]).astype(np.float32)
            codes, pis, zs = dist.all_gather_tuples(codes, pis, zs, consumer=None if self.keep_replica_buffers else 0)   # THE exchange of the round (RCCL all-gather)
            self.last_gathered = dist.last_gather_total
            if self.rank != 0 and not self.keep_replica_buffers:
                return
            states = self.engine.pool.codes_to_planes(codes, 9)
        self.data_buffer.extend(get_equi_data(list(zip(states, pis, zs)), self.board_height, self.board_width))

    def collect_selfplay_data(self, training_index):
        """train_mxnet.py:137-154: SGF replay phase."""
        name = self._training_data[training_index % len(self._training_data)]
        warning, winner, data = self.game.start_self_play(_NoPlayer(), sgf_home=self._sgf_home, file_name=name)
        if warning:
            _logger.error("WARNING: bad SGF record %s", name)
            return
        data = list(data)
        self.episode_len = len(data)
        self.data_buffer.extend(get_equi_data(data, self.board_height, self.board_width))

    # ---- update / evaluation ---------------------------------------------------------------
    def _trainer(self):
        net = self.policy_value_net
        if self._trainer_override is not None:
            return self._trainer_override
        from .train import HipTrainer
        if getattr(net, "_trainer", None) is None:
            # a re-created trainer (set_params dropped the old one) continues the dropout mask sequence instead of replaying it
            net._trainer = HipTrainer(net.params(), net.net_kind, net._n_blocks, batch_size=self.batch_size,
                                      device_index=net._device, seed=self._seed, dropout_step0=self._train_steps)
        return net._trainer

    def policy_update(self, trainer=None, kl_net=None):
        """train_mxnet.py:194-240 (rank 0 of a multi-rank run; see `_exchange_update` / `_trainer_main`).  Lock step: the
        self-play evaluator itself supplies the old / new predictions of the KL monitor (re-folded per epoch);
        asynchronous: the trainer thread's own evaluator `kl_net` does, and the self-play evaluator is left alone."""
        mini = self.data_buffer.sample(self._rng, self.batch_size)
        net = self.policy_value_net if kl_net is None else kl_net
        trainer = self._trainer() if trainer is None else trainer
        t_before = getattr(trainer, "t", 0)
        mon = {}
        loss, entropy, kl, self.lr_multiplier = policy_update(trainer, mini, self.learn_rate, self.lr_multiplier,
                                                              self.epochs, self.kl_targ, evaluator=_KeepTrainer(net), monitors=mon)
        self._train_steps += max(0, getattr(trainer, "t", 0) - t_before)
        self.last_monitors = mon                          # train_mxnet.py:222-239: the value head's explained variance
        _logger.info("kl:%.4f lr:%.1e lr_multiplier:%.3f loss:%.4f entropy:%.4f explained_var_old:%.3f explained_var_new:%.3f",
                     kl, mon["learn_rate"], self.lr_multiplier, loss, entropy, mon["explained_var_old"], mon["explained_var_new"])
        return loss, entropy, kl

    def _exchange_update(self, rec):
        """The update step of one game batch across ranks: rank 0 decides and trains, everybody learns the outcome and, if
        the weights changed, receives them (module docstring).  Single rank: just the update."""
        do = self.rank == 0 and len(self.data_buffer) > self.batch_size
        if not self.distributed:
            if do:
                rec["loss"], rec["entropy"], rec["kl"] = self.policy_update()
            return
        head = [0.0, 0.0, 0.0, 0.0, self.lr_multiplier]
        if do:
            loss, entropy, kl = self.policy_update()
            head = [1.0, loss, entropy, kl, self.lr_multiplier]
        head = dist.broadcast_floats(head, src=0)
        if head[0] == 0.0:
            return
        rec["loss"], rec["entropy"], rec["kl"] = head[1], head[2], head[3]
        rec["lr_multiplier"] = head[4]          # informational on ranks > 0: the adaptive state lives on rank 0
        net = self.policy_value_net
        if self.rank == 0:
            tr = self._trainer()
            src = tr.p if hasattr(tr, "p") else tr.get_params()
        else:
            src = self._param_template()
        got = dist.broadcast_params({k: src[k] for k in sorted(src)}, src=0)
        if self.rank != 0:
            first = next(iter(got.values()))
            if getattr(first, "is_cuda", False) and hasattr(net, "load_device_params"):
                self._held = got                 # the evaluator packs from these tensors asynchronously: keep them alive
                net.load_device_params(got)
            else:
                net.set_params({k: v.cpu().numpy() for k, v in got.items()})
        self.weight_broadcasts = getattr(self, "weight_broadcasts", 0) + 1

    def _param_template(self):
        """Names and shapes of the parameter set (ranks > 0 never build a trainer; values are overwritten by the broadcast)."""
        if getattr(self, "_template", None) is None:
            self._template = {k: np.zeros(v.shape, np.float32) for k, v in self.policy_value_net.params().items()}
        return self._template

    def policy_evaluate(self, n_games=None, net=None):
        """train_mxnet.py:242-263: win ratio of the current net against pure MCTS."""
        res = Arena(net or self.policy_value_net, self.board_width, self.board_height, self.n_in_row,
                    n_playout=self.n_playout, c_puct=self.c_puct, pure_mcts_playout_num=self.pure_mcts_playout_num,
                    base_seed=len(self.history)).play(n_games or self.eval_games)
        return win_ratio(res)

    def run(self):
        """train_mxnet.py:265-300; with several ranks every rank runs this loop (module docstring)."""
        if self.async_update:
            return self._run_async()
        lead = self.rank == 0
        for i in range(self.game_batch_num):
            if i < self.sgf_batches and self._training_data:
                if lead:                                    # game records are rank 0's (lock step: the other ranks wait in the next collective)
                    self.collect_selfplay_data(i)
            else:
                self.collect_selfplay_data_ai(self.play_batch_size)
            rec = {"batch": i + 1, "episode_len": self.episode_len, "buffer": len(self.data_buffer)}
            self._exchange_update(rec)
            if lead and (i + 1) % 50 == 0:
                os.makedirs(self.model_dir, exist_ok=True)
                self.policy_value_net.save_model(os.path.join(self.model_dir, "current_policy.model"))
            if lead and (i + 1) % self.check_freq == 0:
                rec["win_ratio"] = wr = self.policy_evaluate()
                if wr > self.best_win_ratio:
                    self.best_win_ratio = wr
                    os.makedirs(self.model_dir, exist_ok=True)
                    self.policy_value_net.save_model(os.path.join(self.model_dir, "best_policy_%s.model" % i))
                    if self.best_win_ratio >= 0.98 and self.pure_mcts_playout_num < 8000:
                        self.pure_mcts_playout_num += 1000
                        self.best_win_ratio = 0.0
            self.history.append(rec)
            _logger.info("batch %s", rec)
        return self.history

    # ---- the asynchronous schedule (module docstring) ---------------------------------------------
    def _play_round(self):
        if self.round_steps:
            self.engine.run_steps(int(self.round_steps))
            return
        t_end = time.perf_counter() + self.round_seconds
        while True:
            self.engine.run_steps(8)
            if time.perf_counter() >= t_end:
                return

    def _schedule_after_batch(self, i, rec, net, first=None):
        """The reference's per-batch schedule behind the update (train_mxnet.py:283-298): checkpoint every 50 batches,
        arena every check_freq.  i = 0-based game-batch index.  `first` (asynchronous schedule): the update covered the
        game batches first ... i (a cluster of games finished in one round); the weights are the same for all of them, so a
        checkpoint / an arena evaluation runs ONCE if the range crosses a multiple, not once per skipped index."""
        lo = i if first is None else first
        crosses = lambda m: (i + 1) // m > lo // m          # some k in (lo, i + 1] is a multiple of m
        if crosses(50):
            os.makedirs(self.model_dir, exist_ok=True)
            net.save_model(os.path.join(self.model_dir, "current_policy.model"))
        if crosses(self.check_freq):
            rec["win_ratio"] = wr = self.policy_evaluate(net=net)
            if wr > self.best_win_ratio:
                self.best_win_ratio = wr
                os.makedirs(self.model_dir, exist_ok=True)
                net.save_model(os.path.join(self.model_dir, "best_policy_%s.model" % i))
                if self.best_win_ratio >= 0.98 and self.pure_mcts_playout_num < 8000:
                    self.pure_mcts_playout_num += 1000
                    self.best_win_ratio = 0.0

    def _trainer_main(self):
        """Rank 0's trainer thread: replay buffer, policy_update, checkpoints, arena -- never on the self-play path."""
        try:
            import contextlib
            ctx = contextlib.nullcontext()
            net = self.policy_value_net
            kl_net = self._eval_net
            trainer = self._trainer_override
            if trainer is None:
                import torch
                from .train import HipTrainer
                stream = torch.cuda.Stream(device=net._device)
                ctx = torch.cuda.stream(stream)      # this thread's kernels: a stream of their own, beside the self-play stream
                if kl_net is None:
                    kl_net = self._eval_net = PolicyValueNet(
                        self.board_width, self.board_height, self.batch_size, n_blocks=net._n_blocks,
                        n_filter=net._n_filter, model_params=net.params(), net_kind=net.net_kind, device=net._device)
                    self._own_eval_net = True
            elif kl_net is None:
                kl_net = self._eval_net = net
            with ctx:
                if trainer is None:
                    trainer = HipTrainer(net.params(), net.net_kind, net._n_blocks, batch_size=self.batch_size,
                                         device_index=net._device, seed=self._seed)
                self._async_trainer = trainer
                games_recv = 0
                busy_until = 0.0
                stop = False
                # SGF bootstrap (train_mxnet.py:270-271: the first batches replay game records instead of searching):
                # game batches 0 ... n_sgf - 1, one record + one policy_update each, here on the trainer thread.  Self-play
                # starts from the net these batches trained (train_mxnet.py:268-272): every rank HOLDS in _run_async until
                # `_sgf_done` is set, so no game of the untrained net enters the buffer or counts toward the budget.
                n_sgf = self._sgf_phase_batches()
                for i in range(n_sgf):
                    before = len(self.data_buffer)
                    self.collect_selfplay_data(i)
                    # (the record's own length: self.episode_len belongs to the main thread once self-play runs)
                    ep_len = (len(self.data_buffer) - before) // 8 if len(self.data_buffer) > before else 0
                    rec = {"batch": i + 1, "sgf": True, "episode_len": ep_len or self.episode_len, "buffer": len(self.data_buffer)}
                    if len(self.data_buffer) == before:
                        rec["skipped"] = True                            # an unreadable record: nothing joined the buffer
                    if len(self.data_buffer) > self.batch_size:
                        t0 = time.time()
                        loss, entropy, kl = self.policy_update(trainer, kl_net)
                        snap = self._snapshot(trainer)
                        self.update_intervals.append((t0, time.time()))
                        with self._lock:
                            self.updates_done += 1
                            self._last = (loss, entropy, kl)
                            self._fresh = (self.updates_done, snap)
                        rec.update(loss=loss, entropy=entropy, kl=kl)
                    self._schedule_after_batch(i, rec, kl_net)
                    with self._lock:
                        self.trainer_history.append(rec)
                self._sgf_done.set()
                batches_done = n_sgf
                while not stop:
                    items = [self._train_q.get()]
                    while True:
                        try:
                            items.append(self._train_q.get_nowait())
                        except queue.Empty:
                            break
                    for it in items:
                        if it is None:
                            stop = True
                            continue
                        codes, pis, zs, n_games = it
                        states = self.engine.pool.codes_to_planes(codes, 9)
                        self.data_buffer.extend(get_equi_data(list(zip(states, pis, zs)), self.board_height, self.board_width))
                        games_recv += n_games
                    due = n_sgf + games_recv // self.play_batch_size - batches_done
                    if due <= 0:
                        continue
                    first = batches_done
                    batches_done += due
                    rec = {"batch": batches_done, "buffer": len(self.data_buffer)}
                    if len(self.data_buffer) > self.batch_size:
                        now = time.time()
                        if now < busy_until and not stop:
                            time.sleep(busy_until - now)            # max_update_share: leave the device to self-play
                        t0 = time.time()
                        if self.exclusive_updates:
                            self._gpu_gate.clear()                  # the self-play loop of this rank waits at its next round
                        try:
                            loss, entropy, kl = self.policy_update(trainer, kl_net)
                            snap = self._snapshot(trainer)
                        finally:
                            self._gpu_gate.set()
                        t1 = time.time()
                        self.update_intervals.append((t0, t1))
                        if self.max_update_share < 1.0:
                            busy_until = t1 + (t1 - t0) * (1.0 / max(self.max_update_share, 1e-3) - 1.0)
                        with self._lock:
                            self.updates_done += 1
                            self.updates_skipped += due - 1
                            self._last = (loss, entropy, kl)
                            self._fresh = (self.updates_done, snap)
                        rec.update(loss=loss, entropy=entropy, kl=kl)
                    self._schedule_after_batch(batches_done - 1, rec, kl_net, first=first)
                    with self._lock:
                        self.trainer_history.append(rec)
        except BaseException as e:          # surfaces in the main thread at the next round (or in the bootstrap hold)
            self._trainer_error = e
            self._sgf_done.set()

    def _sgf_phase_batches(self):
        """How many leading game batches replay SGF records (0 without records), never more than game_batch_num."""
        return min(int(self.sgf_batches), int(self.game_batch_num)) if self._training_data else 0

    def _snapshot(self, trainer):
        """A private copy of the trainer's weights for the main thread to broadcast / load while the trainer goes on."""
        if hasattr(trainer, "p") and hasattr(trainer, "torch"):
            snap = {k: v.clone() for k, v in trainer.p.items()}
            trainer.torch.cuda.current_stream(trainer.device).synchronize()
            return snap
        return trainer.get_params()

    def _install_weights(self, version, snap):
        """One round's weight exchange: rank 0 sends snapshot `snap`, everybody (rank 0 included) loads it into the
        self-play evaluator."""
        net = self.policy_value_net
        src = snap if self.rank == 0 else self._param_template()
        got = dist.broadcast_params({k: src[k] for k in sorted(src)}, src=0) if self.distributed else dict(src)
        first = next(iter(got.values()))
        if getattr(first, "is_cuda", False) and hasattr(net, "load_device_params"):
            self._held = got                 # the evaluator packs from these tensors asynchronously: keep them alive
            net.load_device_params(got)
        else:
            net.set_params({k: (v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)) for k, v in got.items()},
                           **({"_keep_trainer": True} if self._custom_net else {}))
        self.weights_version = version
        self.weight_broadcasts += 1

    def _run_async(self):
        lead = self.rank == 0
        # the SGF bootstrap batches (rank 0's trainer thread) count as game batches, like the reference's loop index
        target_games = (self.game_batch_num - self._sgf_phase_batches()) * self.play_batch_size * self.world
        self.trainer_history = []
        self._games_collected = 0
        if lead:
            self._train_q = queue.Queue()
            self.engine.gate = self._gpu_gate if self.exclusive_updates else None
            self._trainer_thread = threading.Thread(target=self._trainer_main, name="apz-trainer", daemon=True)
            self._trainer_thread.start()
        # The SGF bootstrap first (train_mxnet.py:268-272: self-play starts from the net the game records trained): every
        # rank holds here until rank 0's trainer thread reports the bootstrap batches done.  One short header round trip per
        # interval: no rank sits in a single collective for long, and a trainer failure travels with it.
        bootstrap = self._sgf_phase_batches() > 0
        while True:
            flags = [1.0, 0.0]
            if lead:
                ready = self._sgf_done.wait(timeout=0.5)
                flags = [1.0 if ready else 0.0, 1.0 if self._trainer_error is not None else 0.0]
            if self.distributed:
                flags = dist.broadcast_floats(flags, src=0)
            if flags[1] != 0.0:
                raise RuntimeError("the trainer thread died on rank 0") from (self._trainer_error if lead else None)
            if flags[0] != 0.0:
                break
        rnd = 0
        stop = False
        draining = False            # the budget is reached and rank 0's trainer works off its queue: rounds without play
        while not stop:
            rnd += 1
            if draining:
                time.sleep(0.2)     # (short header rounds instead of one long wait inside a collective: dist's watchdog)
            elif not (bootstrap and rnd == 1):      # round 1 after a bootstrap only installs the bootstrap's weights
                self._play_round()
            eps = [] if draining else self.engine.finished[:]
            del self.engine.finished[:len(eps)]
            self._taken += len(eps)
            if eps:
                self.episode_len = int(np.mean([len(e.moves) for e in eps]))
                codes = np.concatenate([e.codes for e in eps])
                pis = np.concatenate([e.pis for e in eps]).astype(np.float32)
                zs = np.concatenate([e.zs for e in eps]).astype(np.float32)
            else:
                codes = np.zeros((0, self.engine.pool.code_stride), np.uint8)
                pis = np.zeros((0, self.board_width * self.board_height), np.float32)
                zs = np.zeros(0, np.float32)
            n_games = len(eps)
            if self.distributed and not draining:
                n_games = int(round(dist.all_reduce_sum(len(eps))))
                codes, pis, zs = dist.all_gather_tuples(codes, pis, zs, consumer=0)      # THE exchange of the round
            self.last_gathered = n_games
            head = [0.0] * 9
            fresh = None
            failed = False
            if lead:
                self._games_collected += n_games
                if len(zs) and self._trainer_error is None:
                    self._train_q.put((codes, pis, zs, n_games))
                done = self._games_collected >= target_games
                state = 0.0
                if done and self._trainer_error is None:
                    if not draining:
                        self._train_q.put(None)              # drain: the trainer finishes its queue and leaves
                    self._trainer_thread.join(timeout=0.0 if self.distributed else None)
                    state = 2.0 if self._trainer_thread.is_alive() else 1.0
                failed = self._trainer_error is not None        # travels in the header: every rank leaves the loop together
                with self._lock:
                    fresh, self._fresh = self._fresh, None
                    loss, entropy, kl = self._last
                    head = [state, float(fresh[0]) if fresh else float(self.weights_version),
                            float(self.updates_done), loss, entropy, kl, self.lr_multiplier, float(self._games_collected),
                            float(self.updates_skipped)]
                if failed:
                    head[0] = -1.0
            if self.distributed:
                head = dist.broadcast_floats(head, src=0)
            if head[0] < 0.0:
                # rank 0 raises with the cause; the others raise too instead of blocking in the next collective
                raise RuntimeError("the trainer thread died on rank 0") from (self._trainer_error if lead else None)
            version = int(head[1])
            rec = {"round": rnd, "games": n_games, "games_collected": int(head[7]), "version": version,
                   "updates": int(head[2]), "updates_skipped": int(head[8]), "leaf_evals": int(self.engine.stats["leaf_evals"])}
            if version > self.weights_version:
                self._install_weights(version, fresh[1] if fresh else None)
                rec.update(loss=head[3], entropy=head[4], kl=head[5], lr_multiplier=head[6])
            self.round_log.append((time.time(), int(self.engine.stats["leaf_evals"])))
            self.history.append(rec)
            stop = head[0] == 1.0
            draining = head[0] == 2.0
        return self.history

    def close(self):
        self.engine.close()
        self.policy_value_net.close()
        if getattr(self, "_own_eval_net", False):
            self._eval_net.close()
        tr = getattr(self, "_async_trainer", None)
        if tr is not None and tr is not self._trainer_override and hasattr(tr, "close"):
            tr.close()


class _KeepTrainer(object):
    """policy_update's evaluator: the net's own HIP inference engine, re-folded from the trainer's weights without
    dropping the trainer's optimiser state."""

    def __init__(self, net):
        self.net = net

    def policy_value(self, states):
        return self.net.policy_value(states)

    def set_params(self, prm):
        self.net.set_params(prm, _keep_trainer=True)

    def load_device_params(self, tensors, stream=None):
        self.net.load_device_params(tensors, stream)


class _NoPlayer(object):
    def reset_player(self):
        pass
