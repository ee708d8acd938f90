"""ctypes bindings for the two in-tree native libraries.

There is NO fallback: if a library is missing the import error says how to build it.
"""
import ctypes as C
import os

# The tree pool's OpenMP threads must not spin for long between parallel regions: a GPU box grants a CPU quota
# (cgroup cpu.max, e.g. 16 cores), libgomp's default ~1 ms+ of spinning per worker burns it, and the kernel then
# throttles the whole process for the rest of the 100 ms period (measured: a 10 ms stall every ~30-50 self-play
# steps, 6.2 vs 6.8 games/s).  A bounded spin (GOMP_SPINCOUNT=100000) keeps the wake-up latency low for small
# configurations (8x8, 128 games: 0.38 ms/step; OMP_WAIT_POLICY=passive 0.44, default 0.34) and is as fast as
# passive waiting on the large one.  libgomp reads the variable when it initialises, so it is set here, before
# any OpenMP runtime is loaded; an explicit GOMP_SPINCOUNT / OMP_WAIT_POLICY of the caller wins.
if "GOMP_SPINCOUNT" not in os.environ and "OMP_WAIT_POLICY" not in os.environ:
    os.environ["GOMP_SPINCOUNT"] = "100000"

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(PKG, "libalphapig_host.so")
HIP_LIB = os.path.join(PKG, "libalphapig_hip.so")

_host = None
_hip = None


class NativeLibraryMissing(RuntimeError):
    pass


class ApzhConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("n_in_row", C.c_int32),
                ("n_games", C.c_int32), ("n_playout", C.c_int32), ("prior_is_f32", C.c_int32),
                ("n_threads", C.c_int32), ("reserved", C.c_int32), ("c_puct", C.c_double)]


class ApzConfig(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("c_in", C.c_int32),
                ("n_filter", C.c_int32), ("n_blocks", C.c_int32), ("net_kind", C.c_int32),
                ("max_batch", C.c_int32), ("device", C.c_int32)]


def _ptr(t):
    return C.POINTER(t)


def host():
    """libalphapig_host.so (include/alphapig_host.h)."""
    global _host
    if _host is not None:
        return _host
    if not os.path.exists(HOST_LIB):
        raise NativeLibraryMissing(
            "%s not built; run `python -m alphapig_amd.build host` (no pure-Python fallback)" % HOST_LIB)
    L = C.CDLL(HOST_LIB)
    vp, i32p, i16p, i8p, u8p, i64p, f32p, f64p, u32p = (C.c_void_p, _ptr(C.c_int32), _ptr(C.c_int16),
                                                        _ptr(C.c_int8), _ptr(C.c_uint8), _ptr(C.c_int64),
                                                        _ptr(C.c_float), _ptr(C.c_double), _ptr(C.c_uint32))
    sig = {
        "apzh_last_error": (C.c_char_p, []),
        "apzh_version": (C.c_int, []),
        "apzh_create": (vp, [_ptr(ApzhConfig)]),
        "apzh_destroy": (None, [vp]),
        "apzh_game_reset": (C.c_int, [vp, C.c_int, C.c_int]),
        "apzh_game_set_position": (C.c_int, [vp, C.c_int, i16p, i8p, C.c_int, C.c_int]),
        "apzh_game_do_move": (C.c_int, [vp, C.c_int, C.c_int]),
        "apzh_game_status": (C.c_int, [vp, C.c_int, i32p]),
        "apzh_game_history": (C.c_int, [vp, C.c_int, i16p, i8p, C.c_int]),
        "apzh_game_has_a_winner": (C.c_int, [vp, C.c_int, i32p]),
        "apzh_code_stride": (C.c_int, [C.c_int, C.c_int]),
        "apzh_game_codes": (C.c_int, [vp, C.c_int, u8p]),
        "apzh_codes_to_planes": (C.c_int, [u8p, C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
        "apzh_advance": (C.c_int, [vp, i32p, C.c_int, i32p, u8p]),
        "apzh_feed": (C.c_int, [vp, i32p, C.c_int, f32p, f32p]),
        "apzh_feed_advance": (C.c_int, [vp, i32p, C.c_int, f32p, f32p, i32p, u8p]),
        "apzh_feed_sparse": (C.c_int, [vp, C.c_int, i32p, f64p, C.c_int, C.c_double, C.c_int]),
        "apzh_pending_path": (C.c_int, [vp, C.c_int, i16p, C.c_int]),
        "apzh_playouts_done": (C.c_int, [vp, C.c_int]),
        "apzh_set_playouts_done": (C.c_int, [vp, C.c_int, C.c_int]),
        "apzh_set_n_playout": (C.c_int, [vp, C.c_int]),
        "apzh_node_children": (C.c_int, [vp, C.c_int, C.c_int, i32p, i64p, f64p, i8p, f64p, i32p, C.c_int,
                                         i64p, f64p]),
        "apzh_set_prior_mode": (C.c_int, [vp, C.c_int]),
        "apzh_root_visits_dense": (C.c_int, [vp, i32p, C.c_int, i32p, i32p]),
        "apzh_update_with_move": (C.c_int, [vp, C.c_int, C.c_int]),
        "apzh_play_move": (C.c_int, [vp, C.c_int, C.c_int, i32p]),
        "apzh_play_moves": (C.c_int, [vp, i32p, C.c_int, i32p, u8p, i32p, i32p]),
        "apzh_stats": (C.c_int, [vp, C.c_int, i64p]),
        "apzh_pool_info": (C.c_int, [vp, i64p]),
        "apzh_pure_get_move": (C.c_int, [vp, C.c_int, u32p, i32p, i32p, i64p, f64p, C.c_int, i32p]),
        "apzh_mt_seed": (C.c_int, [C.c_uint32, u32p, i32p]),
        "apzh_np_sum": (C.c_double, [f64p, C.c_int64]),
        "apzh_root_sample": (C.c_int, [C.c_int, C.c_int, f64p, i32p, i32p, C.c_double, C.c_double, C.c_int, u32p, i32p,
                                       i32p, f64p, f64p, i32p, C.c_int]),
        "apzh_pretouch_limit_gb": (C.c_double, [C.c_double, C.c_int]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _host = L
    return L


HOST_SYMBOLS = ["apzh_last_error", "apzh_version", "apzh_create", "apzh_destroy", "apzh_game_reset",
                "apzh_game_set_position", "apzh_game_do_move", "apzh_game_status", "apzh_game_history",
                "apzh_game_has_a_winner", "apzh_code_stride", "apzh_game_codes", "apzh_codes_to_planes",
                "apzh_advance", "apzh_feed", "apzh_feed_advance", "apzh_feed_sparse", "apzh_pending_path", "apzh_playouts_done",
                "apzh_set_playouts_done", "apzh_set_n_playout", "apzh_node_children", "apzh_set_prior_mode",
                "apzh_root_visits_dense", "apzh_update_with_move", "apzh_play_move", "apzh_play_moves", "apzh_stats",
                "apzh_pool_info", "apzh_pure_get_move", "apzh_mt_seed", "apzh_np_sum", "apzh_root_sample",
                "apzh_pretouch_limit_gb"]

HIP_SYMBOLS = ["apz_last_error", "apz_version", "apz_device_count", "apz_create", "apz_destroy",
               "apz_param_count", "apz_param_name", "apz_param_size", "apz_load_weights", "apz_forward",
               "apz_forward_host", "apz_forward_codes_host", "apz_forward_codes_async", "apz_submit_codes", "apz_wait", "apz_host_alloc",
               "apz_host_free", "apz_encode_planes", "apz_augment8", "apz_sample_moves_host", "apz_sample_moves_keyed_host", "apz_conv3x3_packed_size", "apz_conv3x3_pack",
               "apz_conv3x3_fwd", "apz_conv3x3_wgrad", "apz_wino_packed_size", "apz_wino_pack", "apz_wino_pack_many", "apz_wino_conv",
               "apz_wino_conv_add", "apz_wino_conv_stats", "apz_bn_fwd", "apz_bn_fwd_stats", "apz_bn_bwd", "apz_bn_bwd_splits", "apz_colsum", "apz_adam_step", "apz_wgrad_wino",
               "apz_conv1x1_fwd", "apz_conv1x1_bwd", "apz_conv1x1_bwd2", "apz_fc_fwd", "apz_fc_bwd", "apz_dropout", "apz_pv_loss",
               "apz_layout_convert", "apz_bias_grad", "apz_add", "apz_load_weights_dev",
               "apz_sync", "apz_stream", "apz_set_forward_graphs",
               "apz_device_alloc", "apz_device_free", "apz_memcpy_h2d", "apz_memcpy_d2h",
               "apz_conv3x3_bench", "apz_layer_io", "apz_set_profiling", "apz_kernel_time_ms", "apz_prewarm", "apz_test_select_trunk", "apz_set_trunk_arith", "apz_trunk_overflows"]


def _one_hip_runtime():
    """A process must hold ONE HIP / HSA runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as
    the system's): whichever copy is mapped first serves every later user.  If this library came first it would
    pull in /opt/rocm's copy, a later `import torch` would map the wheel's copy as well (it is loaded by path), the
    second HSA runtime finds no device and torch.cuda.is_available() turns False -- the trainer, which shares device
    memory and streams with torch, could then not run in a process that had evaluated first.  So: when torch is
    installed but not yet imported, map ITS runtime before ours (no `import torch`: that costs seconds)."""
    import sys
    if "torch" in sys.modules or os.environ.get("APZ_NO_HIP_PRELOAD") == "1":     # opt-out switch
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    lib = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(lib):
        return
    # Only a runtime of the SAME major version may stand in for the one libalphapig_hip.so was linked against: with
    # different SONAMEs both would get mapped anyway (the failure this function exists to prevent), or the kernels would
    # silently run on a runtime they were not built for.
    want, have = _needed_hip_soname(HIP_LIB), _soname(lib)
    if want and have and want != have:
        import warnings
        warnings.warn("torch bundles %s but %s needs %s: not preloading torch's HIP runtime (a later `import torch` in this "
                      "process may not see the GPU)" % (have, os.path.basename(HIP_LIB), want))
        return
    C.CDLL(lib, mode=C.RTLD_GLOBAL)
    if os.environ.get("APZ_LOG_HIP_RUNTIME") == "1":
        print("[alphapig_amd] HIP runtime mapped from %s (%s)" % (lib, have), file=sys.stderr)


def _dynamic_entries(path, tag):
    """DT_* string entries (`tag`: 1 = NEEDED, 14 = SONAME) of an ELF64 shared object, parsed by hand (no binutils at run time)."""
    import struct
    out = []
    try:
        with open(path, "rb") as f:
            data = f.read()
        if data[:4] != b"\x7fELF" or data[4] != 2:
            return out
        shoff, = struct.unpack_from("<Q", data, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
        for sec in secs:
            if sec[1] != 6:                                   # SHT_DYNAMIC
                continue
            strtab = secs[sec[6]]                             # sh_link -> .dynstr
            for off in range(sec[4], sec[4] + sec[5], 16):
                d_tag, d_val = struct.unpack_from("<qQ", data, off)
                if d_tag == 0:
                    break
                if d_tag == tag:
                    start = strtab[4] + d_val
                    out.append(data[start:data.index(b"\0", start)].decode())
    except (OSError, struct.error, ValueError, IndexError):
        pass
    return out


def _soname(path):
    e = _dynamic_entries(os.path.realpath(path), 14)
    return e[0] if e else None


def _needed_hip_soname(path):
    for n in _dynamic_entries(path, 1):
        if n.startswith("libamdhip64.so"):
            return n
    return None


def hip():
    """libalphapig_hip.so (include/alphapig_hip.h)."""
    global _hip
    if _hip is not None:
        return _hip
    if not os.path.exists(HIP_LIB):
        raise NativeLibraryMissing(
            "%s not built; run `python -m alphapig_amd.build hip` (the evaluator has no CPU fallback)" % HIP_LIB)
    _one_hip_runtime()
    L = C.CDLL(HIP_LIB)
    vp, f32p, u8p, i32p = C.c_void_p, _ptr(C.c_float), _ptr(C.c_uint8), _ptr(C.c_int32)
    sig = {
        "apz_last_error": (C.c_char_p, []),
        "apz_version": (C.c_int, []),
        "apz_device_count": (C.c_int, []),
        "apz_create": (vp, [_ptr(ApzConfig)]),
        "apz_destroy": (None, [vp]),
        "apz_param_count": (C.c_int, [vp]),
        "apz_param_name": (C.c_char_p, [vp, C.c_int]),
        "apz_param_size": (C.c_int64, [vp, C.c_int]),
        "apz_load_weights": (C.c_int, [vp, _ptr(C.c_char_p), _ptr(f32p), _ptr(C.c_int64), C.c_int]),
        "apz_forward": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp]),
        "apz_forward_host": (C.c_int, [vp, f32p, C.c_int, f32p, f32p]),
        "apz_forward_codes_host": (C.c_int, [vp, u8p, C.c_int, f32p, f32p]),
        "apz_forward_codes_async": (C.c_int, [vp, vp, C.c_int, vp, vp]),
        "apz_submit_codes": (C.c_int, [vp, C.c_int, u8p, C.c_int]),
        "apz_wait": (C.c_int, [vp, C.c_int, f32p, f32p]),
        "apz_host_alloc": (vp, [C.c_int64]),
        "apz_host_free": (None, [vp]),
        "apz_encode_planes": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
        "apz_augment8": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp]),
        "apz_sample_moves_host": (C.c_int, [vp, i32p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_uint64,
                                           f32p, i32p]),
        "apz_sample_moves_keyed_host": (C.c_int, [vp, i32p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_uint64,
                                        C.POINTER(C.c_uint64), f32p, i32p]),
        "apz_conv3x3_packed_size": (C.c_int64, [C.c_int, C.c_int]),
        "apz_conv3x3_pack": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
        "apz_conv3x3_fwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "apz_conv3x3_wgrad": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "apz_wino_packed_size": (C.c_int64, []),
        "apz_wino_pack": (C.c_int, [vp, vp, C.c_int, vp, vp]),
        "apz_wino_pack_many": (C.c_int, [vp, vp, C.c_int, vp, vp]),
        "apz_wino_conv": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
        "apz_bn_fwd": (C.c_int, [vp] * 10 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp]),
        "apz_wino_conv_stats": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, vp]),
        "apz_bn_fwd_stats": (C.c_int, [vp] * 12 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp]),
        "apz_set_forward_graphs": (C.c_int, [vp, C.c_int]),
        "apz_bn_bwd": (C.c_int, [vp] * 13 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "apz_bn_bwd_splits": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "apz_colsum": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_float, vp]),
        "apz_wgrad_wino": (C.c_int, [vp, vp, vp, vp, C.c_int, vp]),
        "apz_adam_step": (C.c_int, [vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
        "apz_wino_conv_add": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
        "apz_conv1x1_bwd2": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "apz_conv1x1_fwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "apz_conv1x1_bwd": (C.c_int, [vp] * 7 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "apz_fc_fwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
        "apz_fc_bwd": (C.c_int, [vp] * 7 + [C.c_int, C.c_int, C.c_int, vp]),
        "apz_dropout": (C.c_int, [vp, vp, vp, C.c_int64, C.c_float, C.c_uint64, C.c_uint64, vp]),
        "apz_pv_loss": (C.c_int, [vp] * 5 + [C.c_int] + [vp] * 6),
        "apz_layout_convert": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, vp]),
        "apz_bias_grad": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
        "apz_add": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
        "apz_load_weights_dev": (C.c_int, [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, vp]),
        "apz_sync": (C.c_int, [vp]),
        "apz_stream": (vp, [vp]),
        "apz_device_alloc": (vp, [vp, C.c_int64]),
        "apz_device_free": (None, [vp, vp]),
        "apz_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_int64]),
        "apz_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_int64]),
        "apz_conv3x3_bench": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
        "apz_layer_io": (C.c_int, [vp, C.c_int, f32p, C.c_int64]),
        "apz_set_profiling": (C.c_int, [vp, C.c_int]),
        "apz_kernel_time_ms": (C.c_int, [vp, C.c_int, f32p]),
        "apz_prewarm": (C.c_int, [vp, C.c_int, C.c_int]),
        "apz_test_select_trunk": (C.c_int, [vp, C.c_int]),
        "apz_set_trunk_arith": (C.c_int, [vp, C.c_int]),
        "apz_trunk_overflows": (C.c_long, [vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _hip = L
    return L


_ARR0 = {}


def as_ptr(arr, ctype):
    """ndarray -> what a POINTER(ctype) parameter accepts.  A zero-length ctypes array over the ndarray's buffer (0.3 us;
    ctypes passes an array instance as the pointer to its first element) instead of arr.ctypes.data_as(...) (2.3 us): the
    self-play scheduler makes five such conversions per group and round, which at BASELINE config 2's 150 us rounds was a
    seventh of the host time.  Read-only arrays (the buffer protocol refuses them) take the slow way."""
    t = _ARR0.get(ctype)
    if t is None:
        t = _ARR0[ctype] = ctype * 0
    try:
        return t.from_buffer(arr)
    except (TypeError, ValueError):
        return arr.ctypes.data_as(C.POINTER(ctype))


def check_c(arr, dtype):
    if not (isinstance(arr, np.ndarray) and arr.dtype == dtype and arr.flags["C_CONTIGUOUS"]):
        raise TypeError("expected C-contiguous %s array" % np.dtype(dtype).name)
    return arr
