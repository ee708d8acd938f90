"""GPU parity tests (run with -m gpu on the MI355X): the HIP evaluator, through its C ABI,
against the CPU oracle (oracle/net_ref.py, float64) on the same seeded inputs.

Tolerance: 1e-4 absolute on pre-softmax policy logits and on the pre-tanh value (north_star),
2e-5 on probabilities / values; exact for the integer kernels (plane encoding, augmentation).
The net oracle itself is "parity unpinned" (MXNet absent) -- see oracle/__init__.py.
"""
import os

import numpy as np
import pytest

from alphapig_amd import weights
from oracle import net_ref

pytestmark = pytest.mark.gpu

LOGIT_ATOL = 1e-4


def random_positions(n, w, c=9, seed=1234):
    """SURVEY 8(d) synthetic leaves: k ~ U{0..80} stones, alternating colours, real encoder."""
    from alphapig_amd.treepool import TreePool
    rs = np.random.RandomState(seed)
    pool = TreePool(w, w, 4 if w < 15 else 5, n_games=1, n_playout=1)
    codes = []
    for _ in range(n):
        k = int(rs.randint(0, min(81, w * w)))
        cells = rs.permutation(w * w)[:k]
        pool.set_position(0, cells, [1 + (i % 2) for i in range(k)], 1 + (k % 2))
        codes.append(pool.codes(0))
    codes = np.stack(codes)
    return codes, pool.codes_to_planes(codes, c)


@pytest.fixture(scope="module")
def resnet3():
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", 15, 15, 9, 3, 128, seed=5, style="bench")
    # trunk_arith "f32": these tests compare bits across batch sizes (the default, "auto" = "f16x2", computes batches of more
    # than 32 boards on the fp16 x 2 split kernel and smaller ones on the exact kernel: same accuracy class, different low bits)
    net = PolicyValueNet(15, 15, batch_size=64, n_blocks=3, n_filter=128, model_params=prm, trunk_arith="f32")
    yield net, prm
    net.close()


def test_library_exports_every_declared_symbol():
    import re
    from alphapig_amd import _native
    L = _native.hip()
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "alphapig_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(apz_[a-z0-9_]+)\s*\(", hdr)))
    assert set(declared) == set(_native.HIP_SYMBOLS)
    for s in declared:
        assert hasattr(L, s)
    assert L.apz_device_count() >= 1


def test_param_table_matches_python(resnet3):
    net, prm = resnet3
    table = net.param_table()
    shapes = weights.param_shapes("resnet", 15, 15, 9, 3, 128)
    assert [n for n, _ in table] == list(shapes.keys())
    assert [s for _, s in table] == [int(np.prod(v)) for v in shapes.values()]


@pytest.mark.parametrize("n", [1, 5, 16, 37, 64])
def test_resnet_layers_and_heads(resnet3, n):
    net, prm = resnet3
    _, planes = random_positions(n, 15, seed=100 + n)
    logits, probs, vlog, vals = net.forward_with_logits(planes)
    o_logits, o_probs, o_vlog, o_vals, (o_stem, o_trunk) = net_ref.forward(prm, planes, "resnet", 3, np.float64, True)
    np.testing.assert_allclose(logits, o_logits, rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(vlog, o_vlog[:, 0], rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(probs, o_probs, rtol=0, atol=2e-5)
    np.testing.assert_allclose(vals, o_vals[:, 0], rtol=0, atol=2e-5)
    assert np.abs(probs.sum(axis=1) - 1).max() < 1e-5
    # per-layer: stem and last trunk conv (needs the host-staged path so planes are resident)
    p2, v2 = net.forward_planes(planes)
    np.testing.assert_array_equal(p2, probs)
    np.testing.assert_array_equal(v2, vals)
    np.testing.assert_allclose(net.layer_output(0, n), o_stem, rtol=0, atol=1e-4)
    np.testing.assert_allclose(net.layer_output(6, n), o_trunk, rtol=0, atol=2e-4)


@pytest.mark.parametrize("kind", ["wino3", "wino3-batched", "wino3b", "wino3h"])
def test_resnet_full_depth_10_blocks(kind):
    """The 10-block net (train_mxnet.py:79-91) against the float64 oracle on every Winograd trunk kernel: 24 boards
    take trunk15_wino3s_kernel by default ("wino3"); "wino3-batched" forces trunk15_wino3_kernel, the kernel the
    self-play bench spends 97 % of its GPU time in, onto the same batch; "wino3b" = trunk_arith "bf16x3", the 3 x bf16
    split kernel forced onto it; "wino3h" = trunk_arith "f16x2", the 2 x fp16 split kernel (the default for batches > 32)."""
    prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
    net = _net_with_trunk_kernel(kind, prm, 10, 32)
    _, planes = random_positions(24, 15, seed=7)
    logits, probs, vlog, vals = net.forward_with_logits(planes)
    o_logits, o_probs, o_vlog, o_vals = net_ref.forward(prm, planes, "resnet", 10, np.float64)
    np.testing.assert_allclose(logits, o_logits, rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(vlog, o_vlog[:, 0], rtol=0, atol=LOGIT_ATOL)
    # reference-style fresh init as well (unit BN stats, zero biases)
    prm2 = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=3, style="reference")
    net.set_params(prm2)
    logits2 = net.forward_with_logits(planes)[0]
    np.testing.assert_allclose(logits2, net_ref.forward(prm2, planes, "resnet", 10)[0], rtol=0, atol=LOGIT_ATOL)
    net.close()


def test_bench_launch_shape_10_blocks_512_boards_against_oracle():
    """The bench's launch: 10 blocks, 512 boards in ONE forward = trunk15_wino3_kernel on a duo grid of 256 workgroups x
    2 items (BASELINE configs[2]: two 512-leaf groups per step).  Rows at both ends of the batch, on both sides of the
    middle (the second item of a workgroup starts at pair 128 = board 256) and 26 random rows are held to the float64
    oracle at north_star's 1e-4 (policy_value_net_mxnet.py:70-102); the remaining rows must carry the same bits as
    the same boards evaluated alone in a small launch (which the tests above hold to the oracle as well)."""
    prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
    from alphapig_amd.policy_value_net import PolicyValueNet
    net = PolicyValueNet(15, 15, batch_size=512, n_blocks=10, n_filter=128, model_params=prm, trunk_arith="f32")
    _, planes = random_positions(512, 15, seed=4242)
    logits, probs, vlog, vals = net.forward_with_logits(planes)
    rows = sorted(set([0, 1, 255, 256, 510, 511]) | set(np.random.RandomState(9).permutation(512)[:26].tolist()))
    o_logits, o_probs, o_vlog, o_vals = net_ref.forward(prm, planes[rows], "resnet", 10, np.float64)
    np.testing.assert_allclose(logits[rows], o_logits, rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(vlog[rows], o_vlog[:, 0], rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(probs[rows], o_probs, rtol=0, atol=2e-5)
    np.testing.assert_allclose(vals[rows], o_vals[:, 0], rtol=0, atol=2e-5)
    for lo, n in ((0, 2), (254, 4), (500, 12)):
        small = net.forward_with_logits(planes[lo:lo + n])       # trunk15_wino3s_kernel: the same bits
        np.testing.assert_array_equal(small[0], logits[lo:lo + n])
        np.testing.assert_array_equal(small[2], vlog[lo:lo + n])
    net.close()


def test_resnet_8x8_and_c_in_4():
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", 8, 8, 9, 2, 128, seed=2, style="bench")
    net = PolicyValueNet(8, 8, batch_size=32, n_blocks=2, n_filter=128, model_params=prm)
    _, planes = random_positions(19, 8, seed=9)
    logits = net.forward_with_logits(planes)[0]
    np.testing.assert_allclose(logits, net_ref.forward(prm, planes, "resnet", 2)[0], rtol=0, atol=LOGIT_ATOL)
    net.close()
    # the north_star measurement shape: 4 input planes (game.py:96-115 encoder)
    prm4 = weights.init_params("resnet", 15, 15, 4, 1, 128, seed=4, style="bench")
    net4 = PolicyValueNet(15, 15, batch_size=32, n_blocks=1, n_filter=128, model_params=prm4, c_in=4)
    _, planes4 = random_positions(21, 15, c=4, seed=11)
    logits4 = net4.forward_with_logits(planes4)[0]
    np.testing.assert_allclose(logits4, net_ref.forward(prm4, planes4, "resnet", 1)[0], rtol=0, atol=LOGIT_ATOL)
    net4.close()


def test_simple_net_8x8_config2():
    """BASELINE config 2 evaluator: policy_value_net_mxnet_simple on 8x8."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, 9, seed=6, style="bench")
    net = PolicyValueNet(8, 8, batch_size=64, model_params=prm, net_kind="simple")
    _, planes = random_positions(64, 8, seed=13)
    logits, probs, vlog, vals = net.forward_with_logits(planes)
    o = net_ref.forward(prm, planes, "simple", dtype=np.float64, return_trunk=True)
    np.testing.assert_allclose(logits, o[0], rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(vlog, o[2][:, 0], rtol=0, atol=LOGIT_ATOL)
    net.forward_planes(planes)
    np.testing.assert_allclose(net.layer_output(0, 64), o[4][0], rtol=0, atol=1e-4)
    np.testing.assert_allclose(net.layer_output(5, 64), o[4][1], rtol=0, atol=1e-4)
    net.close()


def test_simple_net_15x15():
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 15, 15, 9, seed=8, style="bench")
    net = PolicyValueNet(15, 15, batch_size=16, model_params=prm, net_kind="simple")
    _, planes = random_positions(9, 15, seed=17)
    logits = net.forward_with_logits(planes)[0]
    np.testing.assert_allclose(logits, net_ref.forward(prm, planes, "simple")[0], rtol=0, atol=LOGIT_ATOL)
    net.close()


def test_encode_planes_on_device_matches_golden(resnet3, golden_dir):
    """codes -> planes on the GPU is the same integer map as Board.current_state."""
    net, prm = resnet3
    codes, planes = random_positions(40, 15, seed=21)
    p_codes, v_codes = net.evaluate_codes(codes)
    p_planes, v_planes = net.forward_planes(planes)
    np.testing.assert_array_equal(p_codes, p_planes)
    np.testing.assert_array_equal(v_codes, v_planes)
    # and against planes the reference itself produced
    from alphapig_amd.game import Board
    g = np.load(os.path.join(golden_dir, "planes.npz"))
    bc, bp = [], []
    for k in range(int(g["n_cases"])):
        w, n, sp, _ = [int(x) for x in g["c%d_meta" % k]]
        if w != 15:
            continue
        b = Board(width=w, height=w, n_in_row=n)
        b.init_board(sp)
        for m in g["c%d_moves" % k]:
            b.do_move(int(m))
        bc.append(b.position_codes())
        bp.append(g["c%d_planes" % k].astype(np.float32))
    pa, va = net.evaluate_codes(np.stack(bc))
    pb, vb = net.forward_planes(np.stack(bp))
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(va, vb)


def test_policy_value_fn_contract(resnet3):
    """Types of the reference boundary (policy_value_net_mxnet.py:274-280)."""
    net, prm = resnet3
    from alphapig_amd.game import Board
    b = Board(width=15, height=15, n_in_row=5)
    b.init_board()
    for m in (112, 113, 97):
        b.do_move(m)
    pairs, v = net.policy_value_fn(b)
    pairs = list(pairs)
    assert [a for a, _ in pairs] == b.availables
    assert all(isinstance(p, np.float32) for _, p in pairs)
    assert isinstance(v, np.ndarray) and v.shape == (1,) and v.dtype == np.float32
    acts, vals = net.policy_value(np.stack([np.ascontiguousarray(b.current_state())] * 3))
    assert acts.shape == (3, 225) and vals.shape == (3, 1) and acts.dtype == np.float32
    ref = net_ref.forward(prm, np.ascontiguousarray(b.current_state())[None], "resnet", 3)
    np.testing.assert_allclose(np.array([p for _, p in pairs]), ref[1][0][b.availables], rtol=0, atol=2e-5)
    np.testing.assert_allclose(v, ref[3][0], rtol=0, atol=2e-5)


def test_batch_larger_than_max_and_errors(resnet3):
    net, prm = resnet3
    _, planes = random_positions(150, 15, seed=31)     # > batch_size 64: chunked
    p, v = net.forward_planes(planes)
    p1, v1 = net.forward_planes(planes[100:101])
    np.testing.assert_array_equal(p[100], p1[0])
    from alphapig_amd.policy_value_net import PolicyValueNet as _PVN
    auto = _PVN(15, 15, batch_size=64, n_blocks=3, n_filter=128, model_params=prm)       # the default arithmetic
    assert auto.trunk_arith == "f16x2"
    pa, va = auto.forward_planes(planes)
    np.testing.assert_allclose(pa, p, rtol=0, atol=2e-6)            # 64-board chunks on the split kernel: the same values
    np.testing.assert_allclose(va, v, rtol=0, atol=2e-6)
    np.testing.assert_array_equal(auto.forward_planes(planes[100:101])[0], p1)   # one board: the exact kernel, the same bits
    assert auto.trunk_overflows() == 0
    auto.close()
    with pytest.raises(ValueError):
        net.forward_planes(np.zeros((2, 9, 8, 8), np.float32))
    from alphapig_amd.policy_value_net import EvaluatorError, PolicyValueNet
    with pytest.raises(EvaluatorError):
        PolicyValueNet(10, 10, batch_size=4, n_blocks=1)
    bad = dict(prm)
    del bad["bnA1_gamma"]
    with pytest.raises(EvaluatorError):
        net.set_params(bad)


def test_augment8_matches_reference_tables(resnet3, golden_dir):
    import ctypes as C
    net, prm = resnet3
    g = np.load(os.path.join(golden_dir, "equi.npz"))
    k = [i for i in range(int(g["n"])) if int(g["e%d_w" % i]) == 15][0]
    state = np.arange(9 * 225, dtype=np.float32).reshape(1, 9, 15, 15)
    pi = np.arange(225, dtype=np.float32).reshape(1, 225)
    L, h = net.L, net._h
    d = [L.apz_device_alloc(h, x) for x in (state.nbytes, pi.nbytes, state.nbytes * 8, pi.nbytes * 8)]
    L.apz_memcpy_h2d(h, d[0], state.ctypes.data, state.nbytes)
    L.apz_memcpy_h2d(h, d[1], pi.ctypes.data, pi.nbytes)
    assert L.apz_augment8(h, d[0], d[1], 1, 9, d[2], d[3]) == 0
    so = np.empty((8, 9, 15, 15), np.float32)
    po = np.empty((8, 225), np.float32)
    L.apz_memcpy_d2h(h, so.ctypes.data, d[2], so.nbytes)
    L.apz_memcpy_d2h(h, po.ctypes.data, d[3], po.nbytes)
    for x in d:
        L.apz_device_free(h, x)
    np.testing.assert_array_equal(so.astype(np.int32), g["e%d_state_perm" % k])
    np.testing.assert_array_equal(po.astype(np.int32), g["e%d_pi_perm" % k])


def test_stream_ordered_slots(resnet3):
    """Two batches queued back to back on the one stream give the same numbers as separate calls,
    from different host threads, and slot misuse is reported."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from alphapig_amd.policy_value_net import EvaluatorError
    net, prm = resnet3
    codes_a, _ = random_positions(33, 15, seed=41)
    codes_b, _ = random_positions(64, 15, seed=42)
    pa, va = net.evaluate_codes(codes_a)
    pb, vb = net.evaluate_codes(codes_b)
    with ThreadPoolExecutor(max_workers=2) as ex:
        for _ in range(5):
            fa = ex.submit(net.evaluate_codes_slot, 0, codes_a)
            fb = ex.submit(net.evaluate_codes_slot, 1, codes_b)
            ra, rb = fa.result(), fb.result()
            np.testing.assert_array_equal(ra[0], pa)
            np.testing.assert_array_equal(ra[1], va)
            np.testing.assert_array_equal(rb[0], pb)
            np.testing.assert_array_equal(rb[1], vb)
    with pytest.raises(EvaluatorError):
        net._ck(net.L.apz_wait(net._h, 2, pa.ctypes.data_as(C.POINTER(C.c_float)),
                               va.ctypes.data_as(C.POINTER(C.c_float))))
    with pytest.raises(EvaluatorError):
        net.evaluate_codes_slot(7, codes_a)


def test_augment8_gpu_wrapper_equals_host(resnet3):
    from alphapig_amd.augment import augment8_gpu, get_equi_data
    net, prm = resnet3
    rs = np.random.RandomState(5)
    planes = (rs.rand(7, 9, 15, 15) > 0.5).astype(np.float32)
    pis = rs.rand(7, 225).astype(np.float32)
    xo, po = augment8_gpu(net, planes, pis)
    ext = get_equi_data([(planes[i], pis[i], 0.0) for i in range(7)], 15, 15)
    np.testing.assert_array_equal(xo, np.stack([e[0] for e in ext]))
    np.testing.assert_array_equal(po, np.stack([e[1] for e in ext]))


def test_empty_and_ragged_batches(resnet3):
    """n = 0 batches are accepted everywhere; ragged sizes around the 16-board head tile and the
    batch limit give the same rows as single evaluations."""
    net, prm = resnet3
    p, v = net.forward_planes(np.zeros((0, 9, 15, 15), np.float32))
    assert p.shape == (0, 225) and v.shape == (0,)
    p, v = net.evaluate_codes(np.zeros((0, net.code_stride), np.uint8))
    assert p.shape == (0, 225) and v.shape == (0,)
    codes, planes = random_positions(64, 15, seed=55)
    ref_p, ref_v = net.forward_planes(planes)
    for n in (1, 15, 16, 17, 31, 33, 63, 64):
        p, v = net.forward_planes(planes[:n])
        np.testing.assert_array_equal(p, ref_p[:n])
        np.testing.assert_array_equal(v, ref_v[:n])
        p, v = net.evaluate_codes_slot(n % 4, codes[:n])
        np.testing.assert_array_equal(p, ref_p[:n])
    # an empty board and a full board are valid inputs
    from alphapig_amd.game import Board
    b = Board(width=15, height=15, n_in_row=5)
    b.init_board()
    pe, ve = net.evaluate_codes(b.position_codes()[None])
    assert abs(pe.sum() - 1) < 1e-5 and -1 <= ve[0] <= 1
    full = np.arange(225)
    rs = np.random.RandomState(2)
    rs.shuffle(full)
    for m in full:
        b.do_move(int(m))
    pf, vf = net.evaluate_codes(b.position_codes()[None])
    ref = net_ref.forward(prm, np.ascontiguousarray(b.current_state())[None], "resnet", 3)
    np.testing.assert_allclose(pf, ref[1], rtol=0, atol=2e-5)


def _net_with_trunk_kernel(kind, prm, n_blocks, batch):
    """kind "ring" = the direct convolution (trunk15_ring_kernel: exact fp32 FMA chains, the in-tree cross-check), "wino3" = the
    fused F(4x4,3x3) Winograd kernel (default); selected through the C ABI's test hook apz_test_select_trunk."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    if kind in ("wino3b", "wino3h"):   # the split kernels (constructor flag), forced onto batches of every size
        net = PolicyValueNet(15, 15, batch_size=batch, n_blocks=n_blocks, n_filter=128, model_params=prm,
                             trunk_arith={"wino3b": "bf16x3", "wino3h": "f16x2"}[kind])
        net._ck(net.L.apz_test_select_trunk(net._h, 4))
        return net
    net = PolicyValueNet(15, 15, batch_size=batch, n_blocks=n_blocks, n_filter=128, model_params=prm, trunk_arith="f32")
    net._ck(net.L.apz_test_select_trunk(net._h, {"ring": 0, "wino3": 3, "wino3-batched": 4}[kind]))
    return net


@pytest.mark.parametrize("n", [1, 2, 7, 33])
def test_trunk_kernels_against_oracle(n):
    """Both trunk kernels (direct, single-pass Winograd pair) against the float64 oracle: layer
    outputs of a plain and of a residual trunk convolution, logits within the path's 1e-4."""
    prm = weights.init_params("resnet", 15, 15, 9, 2, 128, seed=11, style="bench")
    _, planes = random_positions(n, 15, seed=300 + n)
    o_logits, _, o_vlog, _, (o_stem, o_trunk) = net_ref.forward(prm, planes, "resnet", 2, np.float64, True)
    for kind in ("ring", "wino3"):
        net = _net_with_trunk_kernel(kind, prm, 2, 64)
        try:
            logits, _, vlog, _ = net.forward_with_logits(planes)
            np.testing.assert_allclose(logits, o_logits, rtol=0, atol=LOGIT_ATOL, err_msg=kind)
            np.testing.assert_allclose(vlog, o_vlog[:, 0], rtol=0, atol=LOGIT_ATOL, err_msg=kind)
            net.forward_planes(planes)
            np.testing.assert_allclose(net.layer_output(4, n), o_trunk, rtol=0, atol=2e-4, err_msg=kind)
        finally:
            net.close()


def test_trunk_kernels_agree_on_a_large_ragged_batch():
    """1031 boards (odd: the pair kernel's last workgroup holds a single board; > 512: persistent
    workgroups take several boards / pairs): the Winograd kernels against the direct kernel, which the
    other tests pin to the oracle.  Column 15 of the rows16 layout must stay zero (it is the halo)."""
    n = 1031
    prm = weights.init_params("resnet", 15, 15, 9, 1, 128, seed=12, style="bench")
    _, planes = random_positions(n, 15, seed=77)
    outs = {}
    for kind in ("ring", "wino3"):
        net = _net_with_trunk_kernel(kind, prm, 1, n)
        try:
            p, v = net.forward_planes(planes)
            outs[kind] = (p, v, net.layer_output(1, n), net.layer_output(2, n))
        finally:
            net.close()
    for kind in ("wino3",):
        for a, b in zip(outs[kind], outs["ring"]):
            np.testing.assert_allclose(a, b, rtol=0, atol=5e-5, err_msg=kind)
    # the same board gives the same bits wherever it sits in the batch (position-independent kernels)
    for kind in ("wino3",):
        net = _net_with_trunk_kernel(kind, prm, 1, n)
        try:
            perm = np.random.RandomState(0).permutation(n)
            p2, v2 = net.forward_planes(planes[perm])
            np.testing.assert_array_equal(p2, outs[kind][0][perm])
            np.testing.assert_array_equal(v2, outs[kind][1][perm])
        finally:
            net.close()


@pytest.mark.parametrize("n", [1, 2, 7, 32])
def test_small_batch_trunk_kernel_gives_the_batched_kernels_bits(n):
    """Batches of <= 32 boards (policy_value_fn: ONE board, policy_value_net_mxnet.py:261-280) run the trunk on
    trunk15_wino3s_kernel (one board x 16 output channels per workgroup); larger ones on trunk15_wino3_kernel (board pair
    x 64 channels).  Same transform, same accumulation order, same output formulas: the same bits -- here with the
    batched kernel forced onto the small batch through the C ABI's test hook, and against the float64 oracle."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", 15, 15, 9, 3, 128, seed=15, style="bench")
    _, planes = random_positions(n, 15, seed=500 + n)
    small = PolicyValueNet(15, 15, batch_size=64, n_blocks=3, n_filter=128, model_params=prm, trunk_arith="f32")
    batched = PolicyValueNet(15, 15, batch_size=64, n_blocks=3, n_filter=128, model_params=prm, trunk_arith="f32")
    batched._ck(batched.L.apz_test_select_trunk(batched._h, 4))
    a = small.forward_with_logits(planes)
    b = batched.forward_with_logits(planes)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
    small.forward_planes(planes)
    batched.forward_planes(planes)
    for layer in (1, 2, 6):
        np.testing.assert_array_equal(small.layer_output(layer, n), batched.layer_output(layer, n))
    o = net_ref.forward(prm, planes, "resnet", 3, np.float64)
    np.testing.assert_allclose(a[0], o[0], rtol=0, atol=LOGIT_ATOL)
    small.close()
    batched.close()


def test_small_batch_kernel_in_launch_reduction_over_thousands_of_mixed_launches():
    """trunk15_wino3s_kernel's position halves meet through global slabs and one ticket word per (board, channel tile)
    that every launch exchanges its own epoch into (csrc/trunk15_wino3s.h).  3 000 launches of changing batch sizes
    (different subsets of the ticket words in use), residual and plain layers interleaved, on ONE engine: every forward
    must carry the bits the batched kernel gives the same boards (policy_value_net_mxnet.py:77-83 via policy_value)."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", 15, 15, 9, 3, 128, seed=16, style="bench")
    _, planes = random_positions(32, 15, seed=901)
    batched = PolicyValueNet(15, 15, batch_size=32, n_blocks=3, n_filter=128, model_params=prm, trunk_arith="f32")
    batched._ck(batched.L.apz_test_select_trunk(batched._h, 4))
    ref_p, ref_v = batched.forward_planes(planes)
    batched.close()
    small = PolicyValueNet(15, 15, batch_size=32, n_blocks=3, n_filter=128, model_params=prm, trunk_arith="f32")
    sizes = [1, 32, 2, 7, 31, 3, 16, 1, 24, 5]
    rs = np.random.RandomState(4)
    for it in range(500):                                   # 500 forwards x 6 trunk launches
        n = sizes[it % len(sizes)]
        lo = int(rs.randint(0, 32 - n + 1))
        p, v = small.forward_planes(planes[lo:lo + n])
        assert np.array_equal(p, ref_p[lo:lo + n]) and np.array_equal(v, ref_v[lo:lo + n]), (it, n, lo)
    small.close()


def test_prewarm_and_whole_forward_timing_hooks(resnet3):
    """bench.py's measurement hooks through the C ABI: apz_prewarm queues dummy forwards of empty boards without waiting and
    leaves the evaluator's answers alone; with profiling on, EVERY forward is bracketed as a whole (class "forward": the
    numerator of gpu_busy_frac) while the per-kernel classes are sampled every k-th forward."""
    net, prm = resnet3
    _, planes = random_positions(9, 15, seed=321)
    before = net.forward_planes(planes)
    net.prewarm(64, 5)
    net.sync()
    after = net.forward_planes(planes)
    np.testing.assert_array_equal(before[0], after[0])
    np.testing.assert_array_equal(before[1], after[1])
    net.set_profiling(2)
    for _ in range(6):
        net.forward_planes(planes)
    net.prewarm(33, 4)
    net.sync()
    fwd_ms, fwd_n = net.kernel_time_ms("forward")
    trunk_ms, trunk_n = net.kernel_time_ms("trunk")
    net.set_profiling(False)
    assert fwd_n == 10 and fwd_ms > 0                      # six real + four dummy forwards, every one timed
    assert trunk_n == 5 * 6 and 0 < trunk_ms < fwd_ms      # every second forward sampled: 5 forwards x 6 trunk launches
    with pytest.raises(Exception):
        net.prewarm(10 ** 6, 1)                            # larger than max_batch: refused


def test_small_batch_channel_groups_do_not_change_a_boards_bits():
    """The generic 3x3 convolution gives every board of a small batch to several workgroups of 64 output channels
    (launch_conv_t, BASELINE config 2's 32-board launches) and one workgroup per board in large batches; the 1x1 head
    convolution adds channel-slice partial sums in a fixed order.  Same boards, both launch shapes: identical bits
    (policy_value_net_mxnet_simple.py:68-92 evaluated through policy_value, :232-242)."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, 9, seed=6, style="bench")
    net = PolicyValueNet(8, 8, batch_size=1024, model_params=prm, net_kind="simple")
    _, planes = random_positions(1024, 8, seed=21)
    big = net.forward_with_logits(planes)                       # 1024 x 4 channel groups > 2 x CUs: one workgroup per board
    for lo, n in ((0, 32), (100, 7), (512, 64)):
        small = net.forward_with_logits(planes[lo:lo + n])      # grouped launch
        for a, b in zip(small, big):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b)[lo:lo + n])
    o = net_ref.forward(prm, planes[:48], "simple", dtype=np.float64)
    np.testing.assert_allclose(big[0][:48], o[0], rtol=0, atol=LOGIT_ATOL)
    net.close()


def test_wino3_launch_shapes_do_not_change_a_boards_bits():
    """`wino3_grid` gives batches with fewer board pairs than CUs two workgroups per pair (duos padded to whole groups
    of 16 blocks), batches between 256 and 512 boards uneven duos, larger ones one workgroup per CU.  The same boards
    through every shape: identical bits, and the float64 oracle within the path's tolerance
    (policy_value_net_mxnet.py:77-83 through policy_value, :232-242)."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", 15, 15, 9, 2, 128, seed=14, style="bench")
    net = PolicyValueNet(15, 15, batch_size=600, n_blocks=2, n_filter=128, model_params=prm, trunk_arith="f32")
    _, planes = random_positions(600, 15, seed=88)
    big = net.forward_with_logits(planes)                        # 300 pairs: one workgroup per CU
    # 33 ... 128 boards: QUARTER items (four workgroups per pair, 32 output channels each: csrc/trunk15_wino3.h)
    for lo, n in ((0, 1), (3, 2), (10, 17), (50, 33), (7, 64), (60, 101), (300, 128), (40, 130), (100, 300), (0, 512)):
        small = net.forward_with_logits(planes[lo:lo + n])
        for a, b in zip(small, big):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b)[lo:lo + n], err_msg="n=%d" % n)
    o = net_ref.forward(prm, planes[:24], "resnet", 2, np.float64)
    np.testing.assert_allclose(big[0][:24], o[0], rtol=0, atol=LOGIT_ATOL)
    # the quarter items against the float64 oracle directly, and against the 64-channel items forced onto the same batch
    q = net.forward_with_logits(planes[300:364])
    oq = net_ref.forward(prm, planes[300:364], "resnet", 2, np.float64)
    np.testing.assert_allclose(q[0], oq[0], rtol=0, atol=LOGIT_ATOL)
    net._ck(net.L.apz_test_select_trunk(net._h, 5))              # APZ_TRUNK_WINOGRAD_NO_QUARTER
    for a, b in zip(net.forward_with_logits(planes[300:364]), q):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    net.close()


@pytest.mark.parametrize("kind,arith", [("wino3h", "f16x2"), ("wino3b", "bf16x3")])
def test_split_trunk_10_blocks_512_boards_and_launch_shapes(kind, arith):
    """The split trunk kernels -- 2 x fp16 (csrc/trunk15_wino3h.h, trunk_arith "f16x2": the default for batches > 32) and
    3 x bf16 (csrc/trunk15_wino3b.h, "bf16x3") -- at the bench's launch shape: 10 blocks, 512 boards in one forward, rows at
    both ends / around the middle / 26 random ones against the float64 oracle at north_star's 1e-4
    (policy_value_net_mxnet.py:70-102) -- and a board's bits do not depend on the launch shape (1 ... 300 boards, odd
    batches, fewer pairs than CUs), nor on its place in the batch."""
    prm = weights.init_params("resnet", 15, 15, 9, 10, 128, seed=0, style="bench")
    net = _net_with_trunk_kernel(kind, prm, 10, 512)
    _, planes = random_positions(512, 15, seed=4242)
    logits, probs, vlog, vals = net.forward_with_logits(planes)
    rows = sorted(set([0, 1, 255, 256, 510, 511]) | set(np.random.RandomState(9).permutation(512)[:26].tolist()))
    o_logits, o_probs, o_vlog, o_vals = net_ref.forward(prm, planes[rows], "resnet", 10, np.float64)
    np.testing.assert_allclose(logits[rows], o_logits, rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(vlog[rows], o_vlog[:, 0], rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(probs[rows], o_probs, rtol=0, atol=2e-5)
    np.testing.assert_allclose(vals[rows], o_vals[:, 0], rtol=0, atol=2e-5)
    for lo, n in ((0, 1), (3, 2), (10, 17), (40, 131), (100, 300)):
        small = net.forward_with_logits(planes[lo:lo + n])
        for a, b in zip(small, (logits, probs, vlog, vals)):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b)[lo:lo + n], err_msg="n=%d" % n)
    perm = np.random.RandomState(0).permutation(512)
    p2 = net.forward_with_logits(planes[perm])
    np.testing.assert_array_equal(p2[0], logits[perm])
    # without the test hook the constructor flag leaves batches of <= 32 boards on the exact-fp32 small-batch kernel
    from alphapig_amd.policy_value_net import PolicyValueNet
    plain = PolicyValueNet(15, 15, batch_size=64, n_blocks=10, n_filter=128, model_params=prm, trunk_arith="f32")
    flagged = PolicyValueNet(15, 15, batch_size=64, n_blocks=10, n_filter=128, model_params=prm, trunk_arith=arith)
    np.testing.assert_array_equal(plain.forward_with_logits(planes[:8])[0], flagged.forward_with_logits(planes[:8])[0])
    big = flagged.forward_with_logits(planes[:64])[0]
    np.testing.assert_allclose(big, logits[:64], rtol=0, atol=0)            # 64 boards: the split kernel, the same bits as in the 512 batch
    assert net.trunk_overflows() == 0 and flagged.trunk_overflows() == 0
    for x in (net, plain, flagged):
        x.close()


@pytest.mark.parametrize("arith", ["f16x2", "bf16x3"])
def test_split_weights_packed_on_the_device_equal_the_host_pack(arith):
    """The trainer's refresh path (apz_load_weights_dev, policy_value_net_mxnet.py:295-297) packs the split terms of
    U = G g G^T (two fp16 terms of U S[co] and the per-channel scales / three bf16 terms) with a kernel; the constructor
    packs them on the host.  Same double arithmetic, same rounding: the two evaluators give identical bits -- also after
    the weights change."""
    torch = pytest.importorskip("torch")
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", 15, 15, 9, 3, 128, seed=31, style="bench")
    prm2 = weights.init_params("resnet", 15, 15, 9, 3, 128, seed=32, style="bench")
    _, planes = random_positions(70, 15, seed=77)
    host = PolicyValueNet(15, 15, batch_size=128, n_blocks=3, n_filter=128, model_params=prm, trunk_arith=arith)
    dev = PolicyValueNet(15, 15, batch_size=128, n_blocks=3, n_filter=128, model_params=prm2, trunk_arith=arith)
    tensors = {k: torch.tensor(np.ascontiguousarray(v, dtype=np.float32), device="cuda") for k, v in prm.items()}
    dev.load_device_params(tensors)
    a, b = host.forward_with_logits(planes), dev.forward_with_logits(planes)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
    o = net_ref.forward(prm, planes[:16], "resnet", 3, np.float64)
    np.testing.assert_allclose(a[0][:16], o[0], rtol=0, atol=LOGIT_ATOL)
    host.set_params(prm2)                                       # and the host path again, on the other weights
    tensors2 = {k: torch.tensor(np.ascontiguousarray(v, dtype=np.float32), device="cuda") for k, v in prm2.items()}
    dev.load_device_params(tensors2)
    for x, y in zip(host.forward_with_logits(planes), dev.forward_with_logits(planes)):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
    with pytest.raises(Exception):                              # the arithmetic is chosen before the weights are loaded
        host._ck(host.L.apz_set_trunk_arith(host._h, 0))
    host.close()
    dev.close()


@pytest.mark.gpu
def test_f16x2_overflow_repeats_the_forward_on_the_exact_kernel():
    """trunk_arith "f16x2": an activation beyond the fp16 range (|x| > ~655: |V| = |B^T d B| <= 100 |x| overflows the split)
    gives a non-finite trunk output; the kernel raises the forward's word and the entry point that collects the results
    repeats the forward on the exact-fp32 kernel (include/alphapig_hip.h, apz_set_trunk_arith) -- through every entry point
    the Python mirror uses, with the bits of a trunk_arith="f32" evaluator, and never a silent clamp to 0
    (policy_value_net_mxnet.py:77-83: the reference's fp32 convolution has no such range)."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    from alphapig_amd.game import Board
    prm = weights.init_params("resnet", 15, 15, 9, 2, 128, seed=41, style="bench")
    big = dict(prm)
    big["res_conv1_weight"] = np.asarray(prm["res_conv1_weight"], np.float32) * 3.0e4   # stem outputs in the tens of thousands
    _, planes = random_positions(40, 15, seed=8)
    exact = PolicyValueNet(15, 15, batch_size=64, n_blocks=2, n_filter=128, model_params=big, trunk_arith="f32")
    split = PolicyValueNet(15, 15, batch_size=64, n_blocks=2, n_filter=128, model_params=big, trunk_arith="f16x2")
    assert split.trunk_overflows() == 0
    a, b = exact.forward_with_logits(planes), split.forward_with_logits(planes)          # apz_forward
    assert split.trunk_overflows() == 1
    for x, y in zip(a, b):
        assert np.isfinite(np.asarray(y)).all()
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
    pa, pb = exact.forward_planes(planes), split.forward_planes(planes)                 # apz_forward_host
    assert split.trunk_overflows() == 2
    np.testing.assert_array_equal(pa[0], pb[0])
    boards = []
    rs = np.random.RandomState(3)
    for g in range(40):
        bd = Board(width=15, height=15, n_in_row=5)
        bd.init_board(0)
        for m in rs.permutation(225)[:6 + g % 5]:
            bd.do_move(int(m))
        boards.append(bd.position_codes())
    codes = np.stack(boards)
    ca, cb = exact.evaluate_codes(codes), split.evaluate_codes(codes)                   # apz_forward_codes_host
    assert split.trunk_overflows() == 3
    np.testing.assert_array_equal(ca[0], cb[0])
    np.testing.assert_array_equal(ca[1], cb[1])
    sa, sb = exact.evaluate_codes_slot(1, codes), split.evaluate_codes_slot(1, codes)   # apz_submit_codes / apz_wait
    assert split.trunk_overflows() == 4
    np.testing.assert_array_equal(sa[0], sb[0])
    np.testing.assert_array_equal(sa[1], sb[1])
    # ordinary weights afterwards: no repeat, the split kernel's own results
    split.set_params(prm)
    exact.set_params(prm)
    q = split.forward_with_logits(planes)
    assert split.trunk_overflows() == 4
    o = net_ref.forward(prm, planes, "resnet", 2, np.float64)
    np.testing.assert_allclose(q[0], o[0], rtol=0, atol=LOGIT_ATOL)
    exact.close()
    split.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,w,nrow", [("resnet", 15, 5), ("simple", 8, 4)])
def test_forward_graph_replay_is_the_plain_launch_sequence(kind, w, nrow):
    """apz_submit_codes captures the launch sequence of a small batch into a HIP graph on the third submission of a
    (slot, batch size) pair and replays it afterwards: the same kernels with the same arguments, so the same bits as
    plain launches -- across batch sizes and slots, for new positions in the slot's buffer, and after a weight change
    (which drops the graphs: policy_value_net_mxnet.py:244-259 through policy_value_fn / the self-play engine's slots)."""
    from alphapig_amd.game import Board
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params(kind, w, w, 9, 2, 128, seed=21, style="bench")
    net = PolicyValueNet(w, w, batch_size=64, n_blocks=2, n_filter=128, model_params=prm, net_kind=kind)
    rs = np.random.RandomState(4)
    codes = []
    for i in range(40):
        b = Board(width=w, height=w, n_in_row=nrow)
        b.init_board(0)
        for _ in range(rs.randint(0, 12)):
            b.do_move(int(rs.choice(b.availables)))
        codes.append(b.position_codes())
    codes = np.stack(codes)

    def run(slot, lo, n):
        k = net.submit_codes_slot(slot, codes[lo:lo + n])
        p, v = net.wait_slot(slot, k)
        return np.array(p, copy=True), np.array(v, copy=True)

    shapes = [(0, 0, 1), (1, 3, 7), (0, 8, 32), (2, 0, 33)]
    net._ck(net.L.apz_set_forward_graphs(net._h, 0))
    plain = {s: run(*s) for s in shapes}
    plain_b = run(0, 5, 1)                                  # other positions, batch size 1, slot 0
    net._ck(net.L.apz_set_forward_graphs(net._h, 1))
    for rep in range(6):                                    # submissions 1, 2: plain; 3: captured + launched; 4 ..: replayed
        for s in shapes:
            p, v = run(*s)
            np.testing.assert_array_equal(p, plain[s][0])
            np.testing.assert_array_equal(v, plain[s][1])
    p, v = run(0, 5, 1)                                     # the replayed graph reads the slot's buffer: new positions
    np.testing.assert_array_equal(p, plain_b[0])
    np.testing.assert_array_equal(v, plain_b[1])
    prm2 = weights.init_params(kind, w, w, 9, 2, 128, seed=22, style="bench")
    net.set_params(prm2)                                    # graphs dropped; the next submissions use the new weights
    fresh = PolicyValueNet(w, w, batch_size=64, n_blocks=2, n_filter=128, model_params=prm2, net_kind=kind)
    fresh._ck(fresh.L.apz_set_forward_graphs(fresh._h, 0))
    for rep in range(5):
        p, v = run(1, 3, 7)
        k = fresh.submit_codes_slot(1, codes[3:10])
        pf, vf = fresh.wait_slot(1, k)
        np.testing.assert_array_equal(p, np.asarray(pf))
        np.testing.assert_array_equal(v, np.asarray(vf))
    assert float(np.abs(p - plain[(1, 3, 7)][0]).max()) > 0
    net.close()
    fresh.close()
