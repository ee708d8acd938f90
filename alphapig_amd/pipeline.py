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
"""
import logging
import os
import random

import numpy as np

from . import sgf
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
    def __init__(self, conf, init_model=None, policy_value_net=None, device=0, seed=0):
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
                                     forced_opening=(self.board_width == 15 and self.board_height == 15))
        self._taken = 0
        self._rng = random.Random(seed)
        self.episode_len = 0
        self.history = []

    # ---- data collection -----------------------------------------------------------------
    def collect_selfplay_data_ai(self, n_games=1):
        """train_mxnet.py:170-180, batched: take the next n finished games from the engine."""
        while len(self.engine.finished) < n_games:           # completion order, like a game queue
            self.engine.run_steps(16)
        eps = self.engine.finished[:n_games]
        del self.engine.finished[:n_games]
        self._taken += n_games
        self.episode_len = int(np.mean([len(e.moves) for e in eps]))
        states, pis, zs = episodes_to_tuples(eps, self.engine.pool)
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
    def policy_update(self):
        """train_mxnet.py:194-240."""
        mini = self.data_buffer.sample(self._rng, self.batch_size)
        net = self.policy_value_net
        from .train import HipTrainer
        if getattr(net, "_trainer", None) is None:
            net._trainer = HipTrainer(net.params(), net.net_kind, net._n_blocks, batch_size=self.batch_size,
                                      device_index=net._device)
        # the self-play evaluator itself supplies the old / new predictions of the KL monitor (re-folded per epoch)
        loss, entropy, kl, self.lr_multiplier = policy_update(net._trainer, mini, self.learn_rate, self.lr_multiplier,
                                                              self.epochs, self.kl_targ, evaluator=_KeepTrainer(net))
        _logger.info("kl:%.4f lr_multiplier:%.3f loss:%.4f entropy:%.4f", kl, self.lr_multiplier, loss, entropy)
        return loss, entropy, kl

    def policy_evaluate(self, n_games=None):
        """train_mxnet.py:242-263: win ratio of the current net against pure MCTS."""
        res = Arena(self.policy_value_net, self.board_width, self.board_height, self.n_in_row,
                    n_playout=self.n_playout, c_puct=self.c_puct, pure_mcts_playout_num=self.pure_mcts_playout_num,
                    base_seed=len(self.history)).play(n_games or self.eval_games)
        return win_ratio(res)

    def run(self):
        """train_mxnet.py:265-300."""
        for i in range(self.game_batch_num):
            if i < self.sgf_batches and self._training_data:
                self.collect_selfplay_data(i)
            else:
                self.collect_selfplay_data_ai(self.play_batch_size)
            rec = {"batch": i + 1, "episode_len": self.episode_len, "buffer": len(self.data_buffer)}
            if len(self.data_buffer) > self.batch_size:
                rec["loss"], rec["entropy"], rec["kl"] = self.policy_update()
            if (i + 1) % 50 == 0:
                os.makedirs(self.model_dir, exist_ok=True)
                self.policy_value_net.save_model(os.path.join(self.model_dir, "current_policy.model"))
            if (i + 1) % self.check_freq == 0:
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

    def close(self):
        self.engine.close()
        self.policy_value_net.close()


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
