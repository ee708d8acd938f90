"""PolicyValueNet: drop-in for the reference evaluator (reference policy_value_net_mxnet.py:19-309
and policy_value_net_mxnet_simple.py) on hand-written gfx950 HIP kernels.

    PolicyValueNet(board_width, board_height, batch_size=512, n_blocks=8, n_filter=128,
                   model_params=None)
      .policy_value_fn(board) -> (zip(legal_moves, probs[legal] float32), value float32[1])
      .policy_value(state_batch) -> (acts float32 [B, H*W], vals float32 [B, 1])

No MXNet, no CPU fallback: construction fails loudly without libalphapig_hip.so or without an
AMD GPU.  Extra (non-reference) entry points used by the batched engine: `evaluate_codes`
(compact leaf codes, planes encoded on device) and `forward_planes`.
"""
import ctypes as C
import threading

import numpy as np

from . import _native, weights
from ._native import ApzConfig, as_ptr

KERNEL_CLASSES = {"stem": 0, "trunk": 1, "head_conv": 2, "head_fc": 3, "encode": 4, "forward": 5}


class EvaluatorError(RuntimeError):
    pass


class PolicyValueNet(object):
    def __init__(self, board_width, board_height, batch_size=512, n_blocks=8, n_filter=128,
                 model_params=None, net_kind="resnet", c_in=9, device=0, seed=0, init_style="reference", trunk_arith="auto"):
        """trunk_arith: the arithmetic of the 128 -> 128 trunk convolutions of the 15x15 / 128-filter residual net on
        batches of more than 32 boards (smaller batches and every other net always compute exact fp32 products):
          "f32"    exact fp32 products on the fp32 matrix pipe (csrc/trunk15_wino3.h): the bits the parity tests rest on;
          "f16x2"  every fp32 operand as two fp16 terms on the fp16 matrix pipe, fp32 accumulation (csrc/trunk15_wino3h.h):
                   the same accuracy class (<= 1e-4 on the logits against the float64 oracle, tests/test_gpu_winograd_
                   numerics.py), different low-order bits, 1.5x faster; an activation beyond the fp16 range makes the
                   engine repeat that forward on the exact kernel (never a silently wrong result);
          "bf16x3" three bf16 terms, six products (csrc/trunk15_wino3b.h): round 4's form, kept for comparison;
          "auto"   (default) "f16x2" where such kernels exist (15x15 / 128-filter residual net; 8x8 boards), else "f32".
        8x8 boards (round 6): "f16x2" runs every convolution with a multiple of 64 input channels on the fp16 matrix pipe with
        split operands (csrc/conv8_split.h), for EVERY batch size: a board's bits do not depend on the batch there."""
        self.L = _native.hip()
        self.board_width, self.board_height = int(board_width), int(board_height)
        self.batchsize = int(batch_size)
        self.channelnum = int(c_in)
        self._n_blocks, self._n_filter = int(n_blocks), int(n_filter)
        self.net_kind = net_kind
        self._device = int(device)
        self.hw = self.board_width * self.board_height
        self.code_stride = (self.hw + 1 + 15) // 16 * 16
        kind = {"resnet": 0, "simple": 1}[net_kind]
        cfg = ApzConfig(self.board_height, self.board_width, self.channelnum, self._n_filter, self._n_blocks,
                        kind, self.batchsize, int(device))
        self._fn_lock = threading.Lock()      # policy_value_fn: one submit + wait pair on the latency slot at a time
        self._h = self.L.apz_create(C.byref(cfg))
        if not self._h:
            raise EvaluatorError("apz_create failed: %s" % self.L.apz_last_error().decode())
        if trunk_arith not in ("auto", "f32", "bf16x3", "f16x2"):
            raise ValueError("trunk_arith must be 'auto', 'f32', 'f16x2' or 'bf16x3'")
        ring15 = net_kind == "resnet" and self.board_width == 15 and self.board_height == 15 and self._n_filter == 128
        small8 = self.board_width == 8 and self.board_height == 8
        if trunk_arith == "auto":
            trunk_arith = "f16x2" if (ring15 or small8) else "f32"
        self.trunk_arith = trunk_arith
        if trunk_arith != "f32":                                     # before the weights are loaded: they are packed for it
            self._ck(self.L.apz_set_trunk_arith(self._h, {"bf16x3": 1, "f16x2": 2}[trunk_arith]))
        if model_params is None:
            model_params = weights.init_params(net_kind, self.board_height, self.board_width, self.channelnum,
                                               self._n_blocks, self._n_filter, seed=seed, style=init_style)
        self.set_params(model_params)

    # ---- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self.L.apz_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise EvaluatorError("%s (code %d)" % (self.L.apz_last_error().decode(), rc))
        return rc

    # ---- parameters
    def param_table(self):
        n = self._ck(self.L.apz_param_count(self._h))
        return [(self.L.apz_param_name(self._h, i).decode(), int(self.L.apz_param_size(self._h, i)))
                for i in range(n)]

    def set_params(self, model_params, _keep_trainer=False):
        """model_params: {name: array} or the reference's (arg_params, aux_params) pair."""
        if not _keep_trainer:
            self._trainer = None          # externally supplied weights restart the optimiser state
        if isinstance(model_params, (tuple, list)) and len(model_params) == 2:
            merged = dict(model_params[0])
            merged.update(model_params[1])
            model_params = merged
        table = self.param_table()
        keep = []
        names = (C.c_char_p * len(table))()
        ptrs = (C.POINTER(C.c_float) * len(table))()
        sizes = (C.c_int64 * len(table))()
        for i, (name, size) in enumerate(table):
            if name not in model_params:
                raise EvaluatorError("missing parameter %s" % name)
            a = np.ascontiguousarray(np.asarray(model_params[name]), dtype=np.float32)
            keep.append(a)
            names[i] = name.encode()
            ptrs[i] = as_ptr(a, C.c_float)
            sizes[i] = a.size
        self._ck(self.L.apz_load_weights(self._h, names, ptrs, sizes, len(table)))
        self._params = {name: keep[i].copy() for i, (name, _) in enumerate(table)}
        self._device_source = None

    def load_device_params(self, tensors, stream=None):
        """Refresh the evaluator from parameters that already live in device memory -- {name: tensor with .data_ptr()
        and .numel()}, e.g. the trainer's after an optimiser step: BatchNorm folding and weight packing run as kernels
        (apz_load_weights_dev), nothing crosses PCIe.  The host-side copy behind params() is fetched lazily."""
        table = self.param_table()
        names = (C.c_char_p * len(table))()
        ptrs = (C.c_void_p * len(table))()
        sizes = (C.c_int64 * len(table))()
        for i, (name, size) in enumerate(table):
            if name not in tensors:
                raise EvaluatorError("missing parameter %s" % name)
            t = tensors[name]
            if not t.is_contiguous():
                raise EvaluatorError("parameter %s is not contiguous" % name)
            names[i] = name.encode()
            ptrs[i] = t.data_ptr()
            sizes[i] = t.numel()
        self._ck(self.L.apz_load_weights_dev(self._h, names, ptrs, sizes, len(table), C.c_void_p(stream)))
        self._device_source = tensors

    def params(self):
        """{name: float32 array} of every parameter and BatchNorm statistic (one flat table)."""
        src = getattr(self, "_device_source", None)
        if src is not None:                # weights came from device tensors: fetch them once
            self._params = {name: src[name].detach().cpu().numpy().astype(np.float32).copy() for name, _ in self.param_table()}
            self._device_source = None
        return dict(self._params)

    def get_policy_param(self):
        """(arg_params, aux_params) like the reference's Module.get_params()
        (policy_value_net_mxnet.py:301-303): BatchNorm moving statistics are the aux states."""
        from . import mxnet_model
        arg, aux = {}, {}
        for k, v in self.params().items():
            (aux if k.endswith(mxnet_model.AUX_SUFFIXES) else arg)[k] = v.copy()
        return arg, aux

    def save_model(self, model_file, fmt="mxnet"):
        """fmt 'mxnet' (default): the reference's file -- pickle protocol 2 of (arg_params, aux_params) dicts of
        mx.nd.NDArray (policy_value_net_mxnet.py:305-309), which human_play_mxnet.py / train_mxnet.py's
        init_model read with set_params(*model_params).  FORMAT UNVERIFIED: no sample .model file exists in the
        reference tree and MXNet is not installable here, so the NDArray byte layout is restated from the MXNet 1.x
        sources and has only been round-tripped through this package's own reader (alphapig_amd/mxnet_model.py,
        tests/test_mxnet_model.py); a file written here has never been opened by MXNet.  fmt 'flat': a plain
        {name: ndarray} pickle -- the format to use when nothing but this package has to read the file.
        weights.load_params reads both."""
        if fmt == "mxnet":
            from . import mxnet_model
            mxnet_model.save_model(self.params(), model_file)
        elif fmt == "flat":
            weights.save_params(self.params(), model_file)
        else:
            raise ValueError("fmt must be 'mxnet' or 'flat'")

    # ---- forward
    def forward_planes(self, planes):
        """planes [n, C, H, W] -> (probs float32 [n, HW], values float32 [n])."""
        x = np.ascontiguousarray(planes, dtype=np.float32)
        if x.ndim != 4 or x.shape[1:] != (self.channelnum, self.board_height, self.board_width):
            raise ValueError("planes must be [n, %d, %d, %d]" % (self.channelnum, self.board_height, self.board_width))
        n = x.shape[0]
        probs = np.empty((n, self.hw), dtype=np.float32)
        vals = np.empty(n, dtype=np.float32)
        for s in range(0, n, self.batchsize):
            k = min(self.batchsize, n - s)
            self._ck(self.L.apz_forward_host(self._h, as_ptr(x[s:s + k], C.c_float), k,
                                             as_ptr(probs[s:s + k], C.c_float), as_ptr(vals[s:s + k], C.c_float)))
        return probs, vals

    def forward_with_logits(self, planes):
        """Test hook: -> (logits, probs, value_logits, values) for one batch <= batch_size."""
        x = np.ascontiguousarray(planes, dtype=np.float32)
        n = x.shape[0]
        nbytes = lambda cnt: int(cnt) * 4
        dev = [self.L.apz_device_alloc(self._h, nbytes(c)) for c in (x.size, n * self.hw, n, n * self.hw, n)]
        if not all(dev):
            raise EvaluatorError(self.L.apz_last_error().decode())
        try:
            self._ck(self.L.apz_memcpy_h2d(self._h, dev[0], x.ctypes.data, x.nbytes))
            self._ck(self.L.apz_forward(self._h, dev[0], n, dev[1], dev[2], dev[3], dev[4]))
            self._ck(self.L.apz_sync(self._h))
            probs = np.empty((n, self.hw), np.float32)
            vals = np.empty(n, np.float32)
            logits = np.empty((n, self.hw), np.float32)
            vlog = np.empty(n, np.float32)
            for d, a in ((dev[1], probs), (dev[2], vals), (dev[3], logits), (dev[4], vlog)):
                self._ck(self.L.apz_memcpy_d2h(self._h, a.ctypes.data, d, a.nbytes))
        finally:
            for d in dev:
                self.L.apz_device_free(self._h, d)
        return logits, probs, vlog, vals

    def layer_output(self, layer, n):
        """Test hook: output activations of conv layer `layer` for the last forward_planes batch."""
        cout = self._n_filter if self.net_kind == "resnet" else weights.SIMPLE_LAYERS[layer][1]
        out = np.empty((n, cout, self.board_height, self.board_width), dtype=np.float32)
        self._ck(self.L.apz_layer_io(self._h, int(layer), as_ptr(out, C.c_float), out.size))
        return out

    def evaluate_codes(self, codes):
        """codes uint8 [n, code_stride] (Board.position_codes / TreePool.advance) ->
        (probs [n, HW], values [n]); planes are built on the GPU."""
        c = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1, self.code_stride)
        n = c.shape[0]
        probs = np.empty((n, self.hw), dtype=np.float32)
        vals = np.empty(n, dtype=np.float32)
        for s in range(0, n, self.batchsize):
            k = min(self.batchsize, n - s)
            self._ck(self.L.apz_forward_codes_host(self._h, as_ptr(c[s:s + k], C.c_uint8), k,
                                                   as_ptr(probs[s:s + k], C.c_float),
                                                   as_ptr(vals[s:s + k], C.c_float)))
        return probs, vals

    # ---- stream-ordered slots: a second batch queued while the first runs (selfplay pipeline).  The engine has
    # APZ_MAX_SLOTS = 4 of them; SelfPlayEngine's pipeline groups use slots 0 .. n_slots - 1, the last one is reserved for
    # policy_value_fn so that a human-play / serving front end and a self-play engine can share one net.
    n_slots = 3
    latency_slot = 3

    def evaluate_codes_slot(self, slot, codes):
        """submit + wait on one slot; safe to call from one host thread per slot."""
        c = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1, self.code_stride)
        n = c.shape[0]
        if n > self.batchsize or n == 0:
            return self.evaluate_codes(c)
        probs = np.empty((n, self.hw), dtype=np.float32)
        vals = np.empty(n, dtype=np.float32)
        self._ck(self.L.apz_submit_codes(self._h, int(slot), as_ptr(c, C.c_uint8), n))
        self._ck(self.L.apz_wait(self._h, int(slot), as_ptr(probs, C.c_float), as_ptr(vals, C.c_float)))
        return probs, vals

    def submit_codes_slot(self, slot, codes):
        """Enqueue one batch on `slot` (apz_submit_codes: copies the codes into the slot's pinned buffer, queues the forward
        and an event on the engine's one stream) and return at once -> n.  Pair with wait_slot."""
        c = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1, self.code_stride)
        n = c.shape[0]
        if n > self.batchsize or n == 0:
            raise EvaluatorError("submit_codes_slot: batch of %d outside [1, %d]" % (n, self.batchsize))
        self._ck(self.L.apz_submit_codes(self._h, int(slot), as_ptr(c, C.c_uint8), n))
        return n

    def wait_slot(self, slot, n):
        """Block until the batch submitted on `slot` is done -> (probs [n, HW], values [n])."""
        probs = np.empty((n, self.hw), dtype=np.float32)
        vals = np.empty(n, dtype=np.float32)
        self._ck(self.L.apz_wait(self._h, int(slot), as_ptr(probs, C.c_float), as_ptr(vals, C.c_float)))
        return probs, vals

    def sample_moves(self, visits, temp=1.0, alpha=0.3, eps=0.25, seed=0, step=0, keys=None):
        """GPU root sampling (opt-in perf mode): visits int32 [g, HW] with -1 for non-children ->
        (pi float32 [g, HW], moves int32 [g]).  keys: optional uint64 [g]; row i's draw then depends on
        (seed, keys[i]) only (not on its row or on `step`)."""
        v = np.ascontiguousarray(visits, dtype=np.int32).reshape(-1, self.hw)
        g = v.shape[0]
        pi = np.empty((g, self.hw), dtype=np.float32)
        mv = np.empty(g, dtype=np.int32)
        kp = None
        if keys is not None:
            k = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1)
            if k.shape[0] != g:
                raise ValueError("one key per row")
            kp = as_ptr(k, C.c_uint64)
        self._ck(self.L.apz_sample_moves_keyed_host(self._h, as_ptr(v, C.c_int32), g, float(temp), float(alpha),
                                                    float(eps), int(seed), int(step), kp, as_ptr(pi, C.c_float),
                                                    as_ptr(mv, C.c_int32)))
        return pi, mv

    # ---- reference API
    def policy_value(self, state_batch):
        """Batched forward (policy_value_net_mxnet.py:232-242): -> (acts [B,HW], vals [B,1])."""
        states = np.asarray(state_batch)
        probs, vals = self.forward_planes(states.reshape(-1, self.channelnum, self.board_height, self.board_width))
        return probs, vals.reshape(-1, 1)

    def policy_value2(self, state_batch):
        return self.policy_value(state_batch)

    def policy_value_fn(self, board):
        """(action, prob) pairs over the legal moves + value of the position for the player to
        move (policy_value_net_mxnet.py:261-280): priors are NOT renormalised over legal moves.
        A board that offers `position_codes()` (alphapig_amd.game.Board) is evaluated through the zero-copy slot: 240
        bytes in, the first convolution builds the planes of `current_state()` itself (same bits as the planes entry
        point, tests/test_gpu_net.py) -- no staging copies around a one-board forward.  Any other Board-like object
        goes through `current_state()` exactly as in the reference."""
        legal = board.availables
        codes = getattr(board, "position_codes", None)
        if codes is not None and getattr(board, "width", None) == self.board_width and \
                getattr(board, "height", None) == self.board_height:
            # apz_wait takes no engine lock (it only waits on the slot's event), so two threads calling policy_value_fn on
            # one net must not interleave their submit / wait pairs on the shared slot: serialise them here
            with self._fn_lock:
                probs, vals = self.evaluate_codes_slot(self.latency_slot, codes()[None])
        else:
            state = np.ascontiguousarray(board.current_state(), dtype=np.float32)
            probs, vals = self.forward_planes(state.reshape(1, self.channelnum, self.board_height, self.board_width))
        return zip(legal, probs[0][legal]), vals[0:1]

    def train_step(self, state_batch, mcts_probs, winner_batch, learning_rate):
        """One optimiser step (policy_value_net_mxnet.py:282-299) -> (loss, entropy), then the new
        weights are re-folded into the HIP evaluator like the reference re-syncs its predict
        modules (:295-297).  Forward, backward and Adam run on the HIP kernels (alphapig_amd/train.py,
        SURVEY.md 8f rank 1); the self-play hot path never touches them."""
        from .train import HipTrainer
        if getattr(self, "_trainer", None) is None:
            self._trainer = HipTrainer(self.params(), self.net_kind, self._n_blocks, batch_size=self.batchsize,
                                       device_index=self._device)
        loss, entropy = self._trainer.train_step(state_batch, mcts_probs, winner_batch, learning_rate)
        self._trainer.sync_evaluator(self)
        return np.array([loss], dtype=np.float32), np.array([entropy], dtype=np.float32)

    # ---- measurement hooks
    def conv_bench(self, layer, n, iters=100, warmup=20):
        ms = np.zeros(1, dtype=np.float32)
        self._ck(self.L.apz_conv3x3_bench(self._h, int(layer), int(n), int(iters), int(warmup), as_ptr(ms, C.c_float)))
        return float(ms[0])

    def set_profiling(self, on):
        """on: False/0 off, True/1 every forward, k > 1 every k-th forward."""
        self._ck(self.L.apz_set_profiling(self._h, int(on)))

    def kernel_time_ms(self, kernel_class):
        out = np.zeros(2, dtype=np.float32)
        self._ck(self.L.apz_kernel_time_ms(self._h, KERNEL_CLASSES[kernel_class], as_ptr(out, C.c_float)))
        return float(out[0]), int(out[1])

    def trunk_overflows(self):
        """Forwards the engine repeated on the exact-fp32 trunk kernel because an activation left the fp16 range
        (trunk_arith "f16x2"; always 0 for the other arithmetics)."""
        return int(self.L.apz_trunk_overflows(self._h))

    def prewarm(self, n, iters):
        """Enqueue `iters` forwards of n empty boards on the engine stream, without waiting (GPU-only warm-up of a
        measurement: clocks, runtime pools; results are never read)."""
        self._ck(self.L.apz_prewarm(self._h, int(n), int(iters)))

    def sync(self):
        self._ck(self.L.apz_sync(self._h))


class LanedEvaluator(object):
    """k PolicyValueNet handles holding the same weights behind ONE slot interface: slot s runs on lane s % k.  Every lane
    is a C-ABI engine of its own -- its own HIP stream, its own activation buffers -- so the forwards of SelfPlayEngine's
    pipeline groups OVERLAP on the GPU instead of queueing on one stream.  For nets whose per-group launches leave most
    of the chip idle and are bound by launch gaps and round trips: BASELINE configs[1] (8x8 simple net, 32-board
    forwards of seven launches each: policy_value_net_mxnet_simple.py:68-92).  The 15x15 / 1024-game configuration
    fills the chip with every launch and keeps one lane (kernels that never overlap: clean per-kernel timings).
    A position's outputs do not depend on the lane: same kernels, same weights, same bits."""

    def __init__(self, lanes):
        if not lanes:
            raise ValueError("at least one lane")
        self.lanes = list(lanes)
        first = self.lanes[0]
        # every lane contributes its own slots: slot s -> lane s % k, lane slot s // k
        self.n_slots = first.n_slots * len(self.lanes)
        self.batchsize, self.hw, self.code_stride = first.batchsize, first.hw, first.code_stride
        for ln in self.lanes[1:]:
            if (ln.batchsize, ln.hw, ln.code_stride) != (self.batchsize, self.hw, self.code_stride):
                raise ValueError("lanes must be evaluators of the same shape")

    @classmethod
    def like(cls, net, n_lanes, **kw):
        """`net` plus n_lanes - 1 more PolicyValueNet handles built from its parameters."""
        extra = [PolicyValueNet(net.board_width, net.board_height, net.batchsize, n_blocks=net._n_blocks,
                                n_filter=net._n_filter, model_params=net.params(), net_kind=net.net_kind, c_in=net.channelnum, device=net._device,
                                trunk_arith=net.trunk_arith, **kw) for _ in range(int(n_lanes) - 1)]
        return cls([net] + extra)

    def _lane(self, slot):
        return self.lanes[int(slot) % len(self.lanes)], int(slot) // len(self.lanes)

    def evaluate_codes(self, codes):
        return self.lanes[0].evaluate_codes(codes)

    def evaluate_codes_slot(self, slot, codes):
        ln, s = self._lane(slot)
        return ln.evaluate_codes_slot(s, codes)

    def submit_codes_slot(self, slot, codes):
        ln, s = self._lane(slot)
        return ln.submit_codes_slot(s, codes)

    def wait_slot(self, slot, n):
        ln, s = self._lane(slot)
        return ln.wait_slot(s, n)

    def sample_moves(self, *a, **kw):
        return self.lanes[0].sample_moves(*a, **kw)

    def __getattr__(self, name):
        # the rest of the evaluator interface (policy_value, policy_value_fn, save_model, net_kind, _n_blocks, _n_filter,
        # _device, board_width, ...: what TrainPipeline and the trainer read) is lane 0's: the lanes hold the same weights
        if name == "lanes":
            raise AttributeError(name)
        return getattr(self.lanes[0], name)

    def set_params(self, model_params, **kw):
        for ln in self.lanes:
            ln.set_params(model_params, **kw)

    def load_device_params(self, tensors, stream=None):
        """The trainer's refresh (device tensors -> every lane's packed weights, no host round trip)."""
        for ln in self.lanes:
            ln.load_device_params(tensors, stream)

    def trunk_overflows(self):
        return sum(ln.trunk_overflows() for ln in self.lanes)

    def params(self):
        return self.lanes[0].params()

    def sync(self):
        for ln in self.lanes:
            ln.sync()

    def close(self):
        for ln in self.lanes:
            ln.close()
