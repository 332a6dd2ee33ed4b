"""Per-kernel parity: every entry point of the C ABI against the NumPy oracle (fp64) on seeded inputs.

Tolerance: the path computes in fp32 (exact fp32 MFMA = k-ordered fmaf chain); against an fp64
oracle the bound used is |err| <= 2e-4 * max|ref| unless stated (north-star gate on logits is 1e-3).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import tf_semantics as S

pytestmark = pytest.mark.gpu


def _ops():
    from tumblr_emotions_amd import ops
    return ops


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def close(got, ref, tol=2e-4):
    got = got.detach().cpu().numpy().astype(np.float64)
    scale = max(1e-6, np.abs(ref).max())
    err = np.abs(got - ref).max()
    assert err <= tol * scale, "max err %.3e vs scale %.3e" % (err, scale)


CONV_CASES = [
    # (N, H, W, Cin, Cout, k, stride)
    (2, 9, 9, 16, 32, 1, 1),
    (3, 14, 14, 24, 64, 3, 1),       # Cin not a multiple of the 16-wide K tile
    (2, 28, 28, 96, 128, 3, 1),
    (4, 7, 7, 832, 624, 1, 1),       # fused Mixed_5c 1x1: several column tiles
    (2, 13, 11, 48, 176, 3, 1),      # ragged spatial size, Cout not a multiple of 32*k
    (1, 8, 8, 8, 200, 1, 1),
    (2, 10, 10, 20, 12, 3, 2),       # stride 2 SAME (pad goes bottom/right)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_with_stats(case):
    ops = _ops()
    N, H, W, Ci, Co, k, s = case
    rng = np.random.RandomState(1)
    x = rng.normal(size=(N, H, W, Ci))
    w = rng.normal(size=(k, k, Ci, Co)) * 0.1
    ref = S.conv2d_same(x, w, s)
    xd, wd = dev(x), dev(w)
    plan = ops.ConvPlan(N, H, W, Ci, Ci, k, k, s, Co, Co, Ci * Co, 1, Co, flags=ops.DS_EPI_STATS)
    M = plan.M
    z = torch.empty(M, Co, device="cuda")
    stats = torch.zeros(2, Co, plan.partials, device="cuda")
    plan.run(ops._p(xd), ops._p(wd), ops._p(z), stats=ops._p(stats))
    torch.cuda.synchronize()
    close(z, ref.reshape(M, Co))
    close(stats[0].sum(1), ref.reshape(M, Co).sum(0), 1e-3)
    close(stats[1].sum(1), (ref.reshape(M, Co) ** 2).sum(0), 1e-3)


def test_conv_stem_7x7_stride2_folded():
    """Conv2d_1a_7x7 (inception_v1.py:63): Cin=3 padded to 4, KW folded into the channel axis."""
    ops = _ops()
    N, H = 2, 32
    rng = np.random.RandomState(2)
    x = rng.uniform(-1, 1, size=(N, H, H, 3))
    w = rng.normal(size=(7, 7, 3, 64)) * 0.1
    ref = S.conv2d_same(x, w, 2)
    x4 = torch.zeros(N, H, H, 4, device="cuda")
    ops.pad_channels(dev(x), 3, x4, 4, N * H * H)
    w4 = np.zeros((7, 7, 4, 64))
    w4[:, :, :3, :] = w
    wd = dev(w4)
    plan = ops.ConvPlan(N, H, H, 28, 4, 7, 1, 2, 64, 64, 28 * 64, 1, 64, fold_cin=4)
    assert (plan.d.OH, plan.d.OW, plan.d.pad_t, plan.d.pad_l) == (16, 16, 2, 2)
    z = torch.empty(plan.M, 64, device="cuda")
    plan.run(ops._p(x4), ops._p(wd), ops._p(z))
    torch.cuda.synchronize()
    close(z, ref.reshape(-1, 64))


@pytest.mark.parametrize("case", [(2, 32, 32, 4), (3, 37, 29, 3), (5, 64, 64, 4), (1, 224, 224, 4)])
def test_conv_stem_packed_rgb(case):
    """ds_conv_stem: the 7x7/2 stem straight from the packed [N, H, W, 3] images (register-direct, 12-byte loads,
    K order built in LDS), with the BatchNorm statistics about a pivot; odd extents exercise both SAME-padding
    sides and the ragged last tile."""
    ops = _ops()
    N, H, W, cs = case
    rng = np.random.RandomState(12)
    x = rng.uniform(-1, 1, size=(N, H, W, 3))
    w = rng.normal(size=(7, 7, 3, 64)) * 0.1
    ref = S.conv2d_same(x, w, 2)
    ws = np.zeros((7, 7, cs, 64))
    ws[:, :, :3] = w
    plan = ops.StemPlan(N, H, W, cs, 64, 64)
    assert plan.M == ref.shape[0] * ref.shape[1] * ref.shape[2]
    z = torch.full((plan.M, 64), float("nan"), device="cuda")
    stats = torch.zeros(2, 64, plan.partials, device="cuda")
    pivot = dev(rng.normal(size=64))
    xd, wd = dev(x), dev(ws)
    plan.run(ops._p(xd), ops._p(wd), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot))
    torch.cuda.synchronize()
    close(z, ref.reshape(plan.M, 64))
    u = ref.reshape(plan.M, 64) - pivot.double().cpu().numpy()
    close(stats[0].sum(1), u.sum(0), 1e-3)
    close(stats[1].sum(1), (u ** 2).sum(0), 1e-3)


@pytest.mark.parametrize("case", [(2, 32, 32, 4), (5, 64, 64, 4), (3, 96, 64, 3), (7, 160, 192, 4), (3, 224, 224, 4), (1, 31, 224, 4),
                                  (2, 20, 40, 4), (1, 64, 36, 3)])
@pytest.mark.parametrize("bf", [False, True], ids=["f32", "bf16"])
def test_stem_with_the_pool_inside_writes_the_window_maxima_of_the_same_z(case, bf):
    """ds_conv_stem_pool (Conv2d_1a_7x7 -> MaxPool_2a_3x3, inception_v1.py:63-67, in one launch): the pooled maxima are BIT
    identical to ds_conv_stem's z pooled by ds_maxpool_fwd (3x3 / 2 SAME: the same MFMA sequence per output pixel; LDS float
    maxima are exact), whatever the split of the batch's row pairs over the workgroups -- image boundaries, the recomputed row
    behind a workgroup's range, the last row pair of an image (two conv rows) -- and the statistics of the FULL map agree with
    the unfused launch up to the grouping of the partial sums, and with the fp64 oracle.  bf16: the same for ds_conv_stem_pool_bf16
    against ds_conv_stem_bf16 (the oracle then convolves the bf16-rounded operands)."""
    ops = _ops()
    from tumblr_emotions_amd import _lib
    lib = _lib.load()
    N, H, W, cs = case
    f_plain, f_pool = (lib.ds_conv_stem_bf16, lib.ds_conv_stem_pool_bf16) if bf else (lib.ds_conv_stem, lib.ds_conv_stem_pool)
    f_parts = lib.ds_conv_stem_bf16_partials if bf else lib.ds_conv_stem_partials
    rng = np.random.RandomState(41 + H)
    x = rng.uniform(-1, 1, size=(N, H, W, 3))
    w = np.zeros((7, 7, cs, 64))
    w[:, :, :3] = rng.normal(size=(7, 7, 3, 64)) * 0.05
    OH, OW = (H + 1) // 2, (W + 1) // 2
    assert bool(lib.ds_conv_stem_pool_supported(H, W)) == (OH % 2 == 0 and OH >= 10 and OW % 4 == 0 and 16 <= OW <= 112)
    if not lib.ds_conv_stem_pool_supported(H, W):
        z = torch.empty(N, OH // 2, OW // 2, 64, device="cuda")
        assert f_pool(ops._p(dev(x)), ops._p(dev(w)), ops._p(z), None, None, N, H, W, cs, 64, 64, None) != 0
        return
    xt, wt = dev(x), dev(w)
    pivot = dev(rng.normal(size=64) * 0.05)
    M = N * OH * OW
    P0, P1 = f_parts(N, OH, OW), lib.ds_conv_stem_pool_partials(N, OH, OW)
    z = torch.empty(M, 64, device="cuda")
    s0 = torch.zeros(2 * 64 * P0, device="cuda")
    _lib.check(f_plain(ops._p(xt), ops._p(wt), ops._p(z), ops._p(s0), ops._p(pivot), N, H, W, cs, 64, 64, ops._stream()), "stem")
    pooled = torch.empty(N, OH // 2, OW // 2, 64, device="cuda")
    am = torch.empty(N, OH // 2, OW // 2, 64, dtype=torch.uint8, device="cuda")
    ops.maxpool_fwd(z.view(N, OH, OW, 64), pooled, am, N, OH, OW, 64, 3, 2, "SAME")
    zmax = torch.full((N * (OH // 2) * (OW // 2) + 2, 64), 3.0, device="cuda")
    s1 = torch.zeros(2 * 64 * P1, device="cuda")
    _lib.check(f_pool(ops._p(xt), ops._p(wt), ops._p(zmax), ops._p(s1), ops._p(pivot), N, H, W, cs, 64, 64, ops._stream()), "stem_pool")
    beta = torch.zeros(64, device="cuda")
    outs = []
    for st, P in ((s0, P0), (s1, P1)):
        mean, rstd, shift = (torch.empty(64, device="cuda") for _ in range(3))
        ops.bn_finalize(st, P, M, 64, beta, 1e-3, 0.9997, mean, rstd, shift, None, None, pivot=pivot)
        outs.append((mean, rstd))
    torch.cuda.synchronize()
    assert float((zmax[-2:] - 3.0).abs().max()) == 0.0
    assert torch.equal(zmax[:-2].view(N, OH // 2, OW // 2, 64), pooled)
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 1e-6 * max(1.0, float(outs[0][0].abs().max()))
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-5 * float(outs[0][1].abs().max())
    zr = (S.conv2d_same(_bf16_round(x), _bf16_round(w[:, :, :3]), 2) if bf else S.conv2d_same(x, w[:, :, :3], 2)).reshape(M, 64)
    close(outs[1][0], zr.mean(0), 1e-4)
    close(outs[1][1], 1.0 / np.sqrt(zr.var(0) + 1e-3), 1e-4)
    close(zmax[:-2], S.max_pool(zr.reshape(N, OH, OW, 64), 3, 2, "SAME").reshape(-1, 64))


@pytest.mark.parametrize("case", [(2, 32, 32, 4), (3, 37, 29, 3), (5, 64, 64, 4), (1, 224, 224, 4)])
def test_conv_stem_packed_rgb_on_bf16_matrix_cores(case):
    """ds_conv_stem_bf16 (the 16-bit configurations' stem: two kernel rows per three v_mfma_f32_32x32x16_bf16, operands rounded
    to bf16 as they are packed): against the fp64 convolution of the bf16-ROUNDED operands the result is fp32-accumulation
    exact (2e-4 of max|ref|: K order, zero slots, padding sides, ragged last tile all right), against the exact convolution
    the bf16 tolerance (1e-2); statistics about a pivot as ds_conv_stem."""
    ops = _ops()
    N, H, W, cs = case
    rng = np.random.RandomState(13)
    x = rng.uniform(-1, 1, size=(N, H, W, 3))
    w = rng.normal(size=(7, 7, 3, 64)) * 0.1
    ref = S.conv2d_same(_bf16_round(x), _bf16_round(w), 2)
    ws = np.zeros((7, 7, cs, 64))
    ws[:, :, :3] = w
    plan = ops.StemPlan(N, H, W, cs, 64, 64, bf16=True)
    z = torch.full((plan.M, 64), float("nan"), device="cuda")
    stats = torch.zeros(2, 64, plan.partials, device="cuda")
    pivot = dev(rng.normal(size=64))
    xd, wd = dev(x), dev(ws)
    plan.run(ops._p(xd), ops._p(wd), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot))
    torch.cuda.synchronize()
    close(z, ref.reshape(plan.M, 64), 2e-4)
    close(z, S.conv2d_same(x, w, 2).reshape(plan.M, 64), 1e-2)
    u = ref.reshape(plan.M, 64) - pivot.double().cpu().numpy()
    close(stats[0].sum(1), u.sum(0), 1e-3)
    close(stats[1].sum(1), (u ** 2).sum(0), 1e-3)


@pytest.mark.parametrize("case", [(2, 14, 14, 24, 64, 3), (2, 7, 7, 192, 384, 3), (3, 9, 9, 32, 176, 1)])
def test_conv_dgrad_reads_hwio_weights_in_place(case):
    ops = _ops()
    N, H, W, Ci, Co, k = case
    rng = np.random.RandomState(3)
    w = rng.normal(size=(k, k, Ci, Co)) * 0.1
    dy = rng.normal(size=(N, H, W, Co))
    ref = S.conv2d_same_bwd_input(dy, w, (N, H, W, Ci), 1)
    # dgrad as a forward conv: reduction over Co, outputs Ci, taps flipped, weights untouched
    plan = ops.ConvPlan(N, H, W, Co, Co, k, k, 1, Ci, Ci, Ci * Co, Co, 1, flip=1)
    dx = torch.empty(plan.M, Ci, device="cuda")
    dyd, wd = dev(dy), dev(w)          # keep alive: the ABI takes raw pointers
    plan.run(ops._p(dyd), ops._p(wd), ops._p(dx))
    torch.cuda.synchronize()
    close(dx, ref.reshape(-1, Ci))


@pytest.mark.parametrize("case", [(2, 7, 7, 48, 128, 3), (4, 7, 7, 832, 624, 1), (2, 5, 5, 12, 15, 3)])
def test_conv_wgrad(case):
    ops = _ops()
    N, H, W, Ci, Co, k = case
    rng = np.random.RandomState(4)
    x = rng.normal(size=(N, H, W, Ci))
    dy = rng.normal(size=(N, H, W, Co))
    ref = S.conv2d_same_bwd_filter(x, dy, (k, k, Ci, Co), 1)
    plan = ops.WgradPlan(N, H, W, Ci, Ci, k, k, 1, Co, Co)
    ws = torch.empty(max(plan.ws_bytes // 4, 1), device="cuda")
    dw = torch.empty(k, k, Ci, Co, device="cuda")
    xd, dyd = dev(x), dev(dy)
    plan.run(ops._p(xd), ops._p(dyd), ops._p(dw), ops._p(ws), plan.ws_bytes)
    torch.cuda.synchronize()
    close(dw, ref, 3e-4)


WGRAD_DIRECT_CASES = [
    # (N, H, W, Cin, Cout, k, stride, ldx_extra): M >= 512 takes the register-direct kernel
    (16, 7, 7, 48, 128, 3, 1, 0),        # 2-wide x 4-wide vectors
    (16, 7, 7, 832, 624, 1, 1, 0),       # Mixed_5c fused 1x1
    (4, 14, 14, 100, 36, 3, 1, 0),       # 4-wide x 2-wide, ragged blocks
    (8, 15, 15, 16, 24, 3, 2, 0),        # stride 2, odd extent
    (8, 9, 9, 15, 33, 3, 1, 0),          # odd channel counts: scalar loads on both sides
    (8, 14, 14, 64, 96, 3, 1, 32),       # x is a channel slice of a wider buffer (ldx > Cin)
    (3, 14, 14, 192, 384, 3, 1, 0),      # M = 588: waves with ragged quarters
    (8192, 1, 1, 300, 64, 1, 1, 0),      # the text tower's transposed MatMul (H = W = 1)
    (600, 1, 1, 52, 15, 1, 1, 0),        # Logits-like: 15 columns
    (2048, 1, 1, 300, 1024, 1, 1, 0),    # wide outputs (the LSTM matrices' shape class)
    (1024, 1, 1, 128, 1028, 1, 1, 0),    # ... with a ragged last column tile
]


@pytest.mark.parametrize("case", WGRAD_DIRECT_CASES)
def test_conv_wgrad_register_direct(case):
    """wgrad_direct_kernel (coalesced k-major operands straight into the MFMA, four waves summed through LDS,
    split-K slabs): Conv2DBackpropFilter against the fp64 oracle."""
    ops = _ops()
    N, H, W, Ci, Co, k, s, extra = case
    rng = np.random.RandomState(6)
    ld = Ci + extra
    xfull = rng.normal(size=(N, H, W, ld))
    x = xfull[..., extra:]
    OH, OW = -(-H // s), -(-W // s)
    dy = rng.normal(size=(N, OH, OW, Co))
    ref = S.conv2d_same_bwd_filter(x, dy, (k, k, Ci, Co), s)
    plan = ops.WgradPlan(N, H, W, Ci, ld, k, k, s, Co, Co)
    ws = torch.empty(max(plan.ws_bytes // 4, 1), device="cuda")
    dw = torch.full((k, k, Ci, Co), float("nan"), device="cuda")
    xd, dyd = dev(xfull), dev(dy)
    plan.run(C.c_void_p(xd.data_ptr() + 4 * extra), ops._p(dyd), ops._p(dw), ops._p(ws), plan.ws_bytes)
    torch.cuda.synchronize()
    close(dw, ref, 3e-4)


def test_gemm_variants_bias_relu_accum_mask_and_unaligned():
    ops = _ops()
    rng = np.random.RandomState(5)
    M, K, N = 37, 52, 15                      # 15 classes: rows are not 16-byte aligned
    a, w, b = rng.normal(size=(M, K)), rng.normal(size=(K, N)), rng.normal(size=N)
    ad, wd, bd = dev(a), dev(w), dev(b)
    out = torch.empty(M, N, device="cuda")
    ops.gemm_plan(M, K, N, K, N, N, flags=ops.DS_EPI_BIAS).run(ops._p(ad), ops._p(wd), ops._p(out), bias=ops._p(bd))
    close(out, a @ w + b)
    # transposed weights: dX = dY * W^T with W [K,N] read in place, ReluGrad mask fused
    dy = rng.normal(size=(M, N))
    act = rng.normal(size=(M, K))
    dx = torch.empty(M, K, device="cuda")
    dyd, actd = dev(dy), dev(act)
    ops.gemm_plan(M, N, K, N, K, N, transposed_w=True, flags=ops.DS_EPI_MASK, ldmask=K).run(
        ops._p(dyd), ops._p(wd), ops._p(dx), mask=ops._p(actd))
    close(dx, (dy @ w.T) * (act > 0))
    # accumulate + relu with a strided output (ldc > N) and strided input (lda > K)
    big_a = rng.normal(size=(M, K + 12))
    prev = rng.normal(size=(M, N + 5))
    pd, bad = dev(prev), dev(big_a)
    ops.gemm_plan(M, K, N, K + 12, N + 5, N, flags=ops.DS_EPI_ACCUM | ops.DS_EPI_RELU).run(
        ops._p(bad), ops._p(wd), ops._p(pd))
    exp = prev.copy()
    exp[:, :N] = np.maximum(prev[:, :N] + big_a[:, :K] @ w, 0)
    torch.cuda.synchronize()
    close(pd, exp)


def test_gemm_lstm_shape():
    ops = _ops()
    rng = np.random.RandomState(6)
    B, H = 64, 128
    h, kern = rng.normal(size=(B, H)), rng.normal(size=(40 + H, 4 * H)) * 0.1
    kd = dev(kern)
    wh_ptr = C.c_void_p(kd.data_ptr() + 40 * 4 * H * 4)            # rows [D:, :] of the TF kernel
    gates = rng.normal(size=(B, 4 * H))
    gd, hd = dev(gates), dev(h)
    ops.gemm_plan(B, H, 4 * H, H, 4 * H, 4 * H, flags=ops.DS_EPI_ACCUM).run(ops._p(hd), wh_ptr, ops._p(gd))
    close(gd, gates + h @ kern[40:])
    dg = rng.normal(size=(B, 4 * H))
    dh, dgd = torch.zeros(B, H, device="cuda"), dev(dg)
    ops.gemm_plan(B, 4 * H, H, 4 * H, H, 4 * H, transposed_w=True).run(ops._p(dgd), wh_ptr, ops._p(dh))
    torch.cuda.synchronize()
    close(dh, dg @ kern[40:].T)


@pytest.mark.parametrize("f4", [False, True, 1, 2], ids=["F2x2", "F4x4", "F4x4-NB1", "F4x4-NB2"])
@pytest.mark.parametrize("case", [(2, 8, 8, 16, 32), (3, 14, 14, 24, 64), (2, 28, 28, 96, 128), (5, 7, 7, 160, 320),
                                  (2, 13, 11, 48, 176), (1, 9, 10, 8, 40), (2, 56, 56, 64, 192), (3, 14, 14, 32, 64),
                                  (1, 4, 4, 16, 16), (9, 5, 6, 16, 48)])
def test_winograd_conv_forward_and_dgrad_match_oracle(case, f4, tuning_lib):
    """ds_conv_wino (fused Winograd F(2x2,3x3), fp32 MFMA) and ds_conv_wino4 (F(4x4,3x3)) against the fp64
    direct-convolution oracle: forward with BatchNorm statistics about a pivot, and the input gradient through the
    flipped / transposed transformed filter; map sizes that are not multiples of the tile (half-empty border tiles),
    Cout not a multiple of 32, a ragged last tile group, one tile per image.  F4x4-NB1 / -NB2 pin the channel blocks per
    workgroup (ds_debug_conv_wino4_set_nb): at these sizes the launch-time model would pick NB = 1 almost everywhere, and
    the NB = 2 instantiations (position 8's accumulators in architectural registers, two epilogue passes) would only run in
    the full-size step tests."""
    ops = _ops()
    from tumblr_emotions_amd import _lib
    nb_pin = 0 if isinstance(f4, bool) else int(f4)
    f4 = bool(f4)
    assert _lib.load().ds_debug_conv_wino4_set_nb(nb_pin) == 0
    try:
        _winograd_case(ops, case, f4)
    finally:
        _lib.load().ds_debug_conv_wino4_set_nb(0)


def _winograd_case(ops, case, f4):
    N, H, W, Ci, Co = case
    step = 16 if f4 else 8                      # channels per K step: the reduction axis must be a multiple
    if Ci % step:
        assert not (f4 and ops.wino4_supported(H, W, Ci, Co))
        pytest.skip("Cin %% %d != 0: the engine uses the implicit GEMM there" % step)
    rng = np.random.RandomState(5)
    x = rng.normal(size=(N, H, W, Ci))
    w = rng.normal(size=(3, 3, Ci, Co)) * 0.1
    ref = S.conv2d_same(x, w, 1)
    xd, wd = dev(x), dev(w)
    plan = ops.WinoPlan(N, H, W, Ci, Ci, Co, Co, flags=ops.DS_EPI_STATS, f4=f4)
    u = torch.empty(plan.u_elems, device="cuda")
    ops.wino_transform_weights(ops._p(wd), u, Ci, Co, dgrad=False, f4=f4)
    M = plan.M
    z = torch.full((M, Co), float("nan"), device="cuda")
    stats = torch.zeros(2, Co, plan.partials, device="cuda")
    pivot = dev(rng.normal(size=Co) * 0.1)
    plan.run(ops._p(xd), ops._p(u), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot))
    torch.cuda.synchronize()
    zz = ref.reshape(M, Co)
    close(z, zz, 2e-4)
    pv = pivot.cpu().numpy().astype(np.float64)
    close(stats[0].sum(1), (zz - pv).sum(0), 2e-3)
    close(stats[1].sum(1), ((zz - pv) ** 2).sum(0), 2e-3)
    # dgrad: correlation over dz with the flipped, channel-transposed filter
    dy = rng.normal(size=ref.shape)
    g = ops.WinoPlan(N, H, W, Co, Co, Ci, Ci + 4, f4=f4)          # strided output rows (ldz > Cout)
    ud = torch.empty(g.u_elems, device="cuda")
    dx = torch.zeros(M, Ci + 4, device="cuda")
    dyd = dev(dy)
    if Co % step == 0:
        ops.wino_transform_weights(ops._p(wd), ud, Ci, Co, dgrad=True, f4=f4)
        g.run(ops._p(dyd), ops._p(ud), ops._p(dx))
        torch.cuda.synchronize()
        dx_ref = S.conv2d_same_bwd_input(dy, w, (N, H, W, Ci), 1).reshape(-1, Ci)
        close(dx[:, :Ci], dx_ref, 2e-4)
        assert float(dx[:, Ci:].abs().max()) == 0.0
        # DS_EPI_BNSUMS: the dgrad also emits, per column, sum(g) and sum(g*y) with g = dx (y > 0), y the forward
        # activation of the layer whose BatchNorm backward consumes dx (same pixel stride as dx)
        yv = np.maximum(rng.normal(size=(M, Ci)), 0.0) * (rng.uniform(size=(M, Ci)) < 0.7)
        yd = dev(np.pad(yv, ((0, 0), (0, 4)), constant_values=5.0))          # positive garbage in the row padding
        P = g.enable_bnsums()
        sums = torch.full((2, Ci, P), float("nan"), device="cuda")
        dx2 = torch.zeros(M, Ci + 4, device="cuda")
        g.run(ops._p(dyd), ops._p(ud), ops._p(dx2), stats=ops._p(sums), ymask=ops._p(yd))
        torch.cuda.synchronize()
        assert torch.equal(dx2, dx)
        gm = dx_ref * (yv > 0)
        close(sums[0].sum(1), gm.sum(0), 2e-3)
        close(sums[1].sum(1), (gm * yv).sum(0), 2e-3)


@pytest.mark.parametrize("case", [(300, 64, 64), (1000, 192, 176), (777, 480, 304), (513, 296, 512), (260, 40, 96),
                                  (4096, 832, 624), (129, 280, 528), (300, 64, 224), (300, 96, 288), (400, 448, 160)])
def test_wide_1x1_kernel_forward_and_dgrad_match_oracle(case, tuning_lib):
    """The wide-tile register-direct kernel for plain 1x1 / GEMM shapes (gemm_wide_kernel), forced on for every
    shape: forward (n-contiguous HWIO weights, BatchNorm statistics about a pivot) and dgrad (the same tensor read
    k-contiguous); K not a multiple of the 16-channel step (296, 280, 40), N not a multiple of the column tile,
    ragged last row group, strided input rows."""
    ops = _ops()
    from tumblr_emotions_amd import _lib
    lib = _lib.load()
    M, K, N = case
    rng = np.random.RandomState(M + K + N)
    x = rng.normal(size=(M, K + 8))[:, :K]                       # row stride K + 8
    w = rng.normal(size=(K, N)) * 0.1
    xd = dev(np.ascontiguousarray(np.pad(x, ((0, 0), (0, 8)), constant_values=np.nan)))   # NaN in the row padding
    xd = torch.nan_to_num(xd, nan=7.0)                            # (finite garbage: the kernel may read it against zero weights)
    wd = dev(w)
    lib.ds_debug_conv_set_wide(2)
    try:
        plan = ops.ConvPlan(M, 1, 1, K, K + 8, 1, 1, 1, N, N, 0, 1, N, flags=ops.DS_EPI_STATS, pad_t=0, pad_l=0, OH=1, OW=1)
        z = torch.full((M, N), float("nan"), device="cuda")
        stats = torch.zeros(2, N, plan.partials, device="cuda")
        pivot = dev(rng.normal(size=N) * 0.2)
        plan.run(ops._p(xd), ops._p(wd), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot))
        torch.cuda.synchronize()
        ref = x @ w
        close(z, ref)
        pv = pivot.cpu().numpy().astype(np.float64)
        close(stats[0].sum(1), (ref - pv).sum(0), 2e-3)
        close(stats[1].sum(1), ((ref - pv) ** 2).sum(0), 2e-3)
        # dgrad: dx[M, K] = dz[M, N] * w^T, w read in place k(=n here)-contiguous
        if N % 8 == 0 and N >= 32:
            dz = rng.normal(size=(M, N))
            g = ops.gemm_plan(M, N, K, N, K, N, transposed_w=True)
            dx = torch.full((M, K), float("nan"), device="cuda")
            dzd = dev(dz)
            g.run(ops._p(dzd), ops._p(wd), ops._p(dx))
            torch.cuda.synchronize()
            close(dx, dz @ w.T)
            # DS_EPI_ACCUM: dx += dz * w^T (the fused 1x1 dgrad adding onto the pool path's gradient)
            g.d.flags = ops.DS_EPI_ACCUM
            prev = rng.normal(size=(M, K))
            dx.copy_(dev(prev))
            g.run(ops._p(dzd), ops._p(wd), ops._p(dx))
            torch.cuda.synchronize()
            close(dx, prev + dz @ w.T)
            # DS_EPI_BNSUMS (+ ACCUM): column sums of g = dx_final (y > 0) and g * y next to the stores
            yv = np.maximum(rng.normal(size=(M, K)), 0.0) * (rng.uniform(size=(M, K)) < 0.7)
            yd = dev(np.pad(yv, ((0, 0), (0, 4)), constant_values=3.0))
            assert lib.ds_conv_igemm_bnsums_supported(C.byref(g.d)) == 1
            P = g.enable_bnsums(K + 4)
            g.d.flags |= ops.DS_EPI_ACCUM
            sums = torch.full((2, K, P), float("nan"), device="cuda")
            dx.copy_(dev(prev))
            g.run(ops._p(dzd), ops._p(wd), ops._p(dx), mask=ops._p(yd), stats=ops._p(sums))
            torch.cuda.synchronize()
            full = prev + dz @ w.T
            close(dx, full)
            gm = full * (yv > 0)
            close(sums[0].sum(1), gm.sum(0), 2e-3)
            close(sums[1].sum(1), (gm * yv).sum(0), 2e-3)
    finally:
        lib.ds_debug_conv_set_wide(1)


@pytest.mark.parametrize("case", [(1000, 176, 192, (64, 160, 176)), (777, 304, 480, (192, 288, 304)), (513, 296, 512, (160, 272, 296)),
                                  (300, 64, 224, (64,)), (4100, 448, 832, (256, 416, 448)), (260, 32, 96, (32,))])
def test_wide_dgrad_applies_batchnorm_backward_on_load(case, tuning_lib):
    """ds_conv_desc.bnb: the wide 1x1 dgrad reads the layer's z and the activation gradient dy (one to three channel
    ranges with their own pixel strides, boundaries multiples of 16, last range ending off the 16-channel step) and forms
    dz = rstd (g - mean g - xhat mean(g xhat)) as it loads.  Against the fp64 formula + GEMM at 2e-4, and BIT-identical
    to ds_bn_bwd_apply followed by the plain dgrad (plain, DS_EPI_ACCUM and DS_EPI_BNSUMS epilogues)."""
    ops = _ops()
    from tumblr_emotions_amd import _lib
    lib = _lib.load()
    M, K, N, ends = case            # K = the layer's output channels (reduction of the dgrad), N = its input channels
    rng = np.random.RandomState(M + K)
    ldz = K + 8
    z = rng.normal(size=(M, K)) * 1.5 + rng.normal(size=K)
    w = rng.normal(size=(N, K)) * 0.1                   # HWIO [1][1][N][K]: the dgrad reads it k-contiguous
    mean, var = z.mean(0), z.var(0)
    rstd = 1.0 / np.sqrt(var + 1e-3)
    beta = rng.normal(size=K) * 0.3
    shift = beta - mean * rstd
    dy = rng.normal(size=(M, K))
    g = dy * (z * rstd + shift > 0)
    xh = (z - mean) * rstd
    a1, a2 = g.mean(0), (g * xh).mean(0)
    dz = rstd * (g - a1 - xh * a2)
    zd = dev(np.pad(z, ((0, 0), (0, 8)), constant_values=5.0))
    parts, keep, c0 = [], [], 0
    for i, c1 in enumerate(ends):                       # each range in its own buffer with its own row stride
        ld = (c1 - c0) + 4 * i
        t = dev(np.pad(dy[:, c0:c1], ((0, 0), (0, ld - (c1 - c0))), constant_values=9.0))
        keep.append(t)
        parts.append((c0, c1, t.data_ptr(), ld))
        c0 = c1
    md, rd, sd = dev(mean), dev(rstd), dev(shift)
    coef = dev(np.stack([a1, a2]))
    wd = dev(w)
    lib.ds_debug_conv_set_wide(2)
    try:
        plan = ops.LayerPlan(ops.DS_CONV_DGRAD, ops.DS_ARITH_F32, 0, 1, M, 1, N, K, 1, 1, ldz, N, 0)
        assert plan.family == ops.DS_FAM_IGEMM
        assert plan.enable_bn_backward_on_load(md, rd, sd, coef, parts)
        dx = torch.full((M, N), float("nan"), device="cuda")
        plan.run(ops._p(zd), ops._p(wd), ops._p(dx))
        torch.cuda.synchronize()
        close(dx, dz @ w.T)
        # the separate pass + plain dgrad on the same inputs: same bits
        segs = ops.make_segments(parts)
        dzd = zd.clone()
        ops.bn_bwd_apply(dzd, segs, M, K, md, rd, sd, coef, dzd, ldz=ldz)
        ref_plan = ops.LayerPlan(ops.DS_CONV_DGRAD, ops.DS_ARITH_F32, 0, 1, M, 1, N, K, 1, 1, ldz, N, 0)
        dx2 = torch.full((M, N), float("nan"), device="cuda")
        ref_plan.run(ops._p(dzd), ops._p(wd), ops._p(dx2))
        torch.cuda.synchronize()
        close(dzd[:, :K], dz)
        assert torch.equal(dx, dx2)
        # with the accumulate + BatchNorm-sums epilogues (the fused block-input dgrad's form)
        if N >= 32 and N % 8 == 0:
            prev = dev(rng.normal(size=(M, N)))
            yv = dev(np.maximum(rng.normal(size=(M, N)), 0.0))
            outs = []
            for pl, xin in ((plan, zd), (ref_plan, dzd)):
                P = pl.enable_bnsums(N)
                assert P > 0
                pl.d.flags |= ops.DS_EPI_ACCUM
                sums = torch.zeros(2, N, P, device="cuda")
                o = prev.clone()
                pl.run(ops._p(xin), ops._p(wd), ops._p(o), mask=ops._p(yv), stats=ops._p(sums))
                torch.cuda.synchronize()
                outs.append((o, sums))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
            close(outs[0][0], prev.cpu().numpy().astype(np.float64) + dz @ w.T)
    finally:
        lib.ds_debug_conv_set_wide(1)


@pytest.mark.parametrize("case", [(2, 14, 14, 64, 96, 1), (3, 9, 10, 40, 72, 1), (2, 28, 28, 192, 176, 1), (2, 8, 8, 16, 32, 3),
                                  (1, 7, 7, 832, 128, 1), (2, 13, 11, 48, 40, 3)])
def test_fp32_products_from_three_bf16_pieces_match_the_oracle(case):
    """ds_conv_f32x3 (fp32 products on the bf16 matrix cores: every operand split into three bf16 pieces, six
    v_mfma_f32_32x32x16_bf16 per eight fp32 MFMAs) against the fp64 oracle at the bound of the fp32 kernels
    (2e-4 of max|ref|), and against the fp32 kernel itself: its error must stay within 1.5x of the fp32 kernel's.
    1x1 and 3x3, Cin not a multiple of 16, Cout not a multiple of 32, strided input and output rows, BatchNorm
    statistics about a pivot; dgrad weights (flipped, transposed); BatchNorm + ReLU on load (1x1)."""
    ops = _ops()
    N, H, W, Ci, Co, k = case
    rng = np.random.RandomState(N * 100 + Ci)
    x = np.maximum(rng.normal(size=(N, H, W, Ci)), 0.0)
    w = rng.normal(size=(k, k, Ci, Co)) * 0.1
    ref = S.conv2d_same(x, w, 1).reshape(-1, Co)
    M = N * H * W
    xp = np.pad(x, ((0, 0), (0, 0), (0, 0), (0, 4)), constant_values=3.0)           # row stride Ci + 4
    xd, wd = dev(xp), dev(w)
    plan = ops.F32x3Plan(N, H, W, Ci, Ci + 4, k, 1, Co, Co + 4, flags=ops.DS_EPI_STATS)
    wb = torch.empty(ops.weights_f32x3_bytes(Ci, Co, k * k, False), dtype=torch.uint8, device="cuda")
    ops.weights_to_f32x3(ops._p(wd), wb, Ci, Co, k * k, False)
    z = torch.full((M, Co + 4), 5.0, device="cuda")
    stats = torch.zeros(2, Co, plan.partials, device="cuda")
    pivot = dev(rng.normal(size=Co) * 0.1)
    plan.run(ops._p(xd), ops._p(wb), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot))
    torch.cuda.synchronize()
    close(z[:, :Co], ref)
    assert float((z[:, Co:] - 5.0).abs().max()) == 0.0
    pv = pivot.cpu().numpy().astype(np.float64)
    close(stats[0].sum(1), (ref - pv).sum(0), 2e-3)
    close(stats[1].sum(1), ((ref - pv) ** 2).sum(0), 2e-3)
    # against the fp32 kernel: the same error class
    direct = ops.ConvPlan(N, H, W, Ci, Ci + 4, k, k, 1, Co, Co, Ci * Co, 1, Co)
    z32 = torch.empty(M, Co, device="cuda")
    direct.run(ops._p(xd), ops._p(wd), ops._p(z32))
    torch.cuda.synchronize()
    e3 = np.abs(z[:, :Co].cpu().numpy().astype(np.float64) - ref).max()
    e32 = np.abs(z32.cpu().numpy().astype(np.float64) - ref).max()
    assert e3 <= 1.5 * e32 + 1e-7 * np.abs(ref).max(), (e3, e32)
    # dgrad: correlation over dz with the flipped, channel-transposed filter
    dz = rng.normal(size=(N, H, W, Co))
    g = ops.F32x3Plan(N, H, W, Co, Co, k, 1, Ci, Ci)
    if Co % 8 == 0:
        wg = torch.empty(ops.weights_f32x3_bytes(Ci, Co, k * k, True), dtype=torch.uint8, device="cuda")
        ops.weights_to_f32x3(ops._p(wd), wg, Ci, Co, k * k, True)
        dx = torch.empty(M, Ci, device="cuda")
        dzd = dev(dz)
        g.run(ops._p(dzd), ops._p(wg), ops._p(dx))
        torch.cuda.synchronize()
        close(dx, S.conv2d_same_bwd_input(dz, w, (N, H, W, Ci), 1).reshape(-1, Ci))
    if k == 1 and Co % 8 == 0:      # the wide kernel's dgrad epilogue: accumulate onto dx, emit the consumer's BatchNorm sums
        yv = np.maximum(rng.normal(size=(M, Ci)), 0.0) * (rng.uniform(size=(M, Ci)) < 0.7)
        prev = rng.normal(size=(M, Ci))
        g.d.flags = ops.DS_EPI_ACCUM | ops.DS_EPI_BNSUMS
        g.d.ldmask = Ci
        P = g.partials_for_sums
        sums = torch.full((2, Ci, P), float("nan"), device="cuda")
        dxa, yd = dev(prev), dev(yv)
        g.run(ops._p(dzd), ops._p(wg), ops._p(dxa), stats=ops._p(sums), mask=ops._p(yd))
        torch.cuda.synchronize()
        want = prev + S.conv2d_same_bwd_input(dz, w, (N, H, W, Ci), 1).reshape(-1, Ci)
        close(dxa, want)
        gm = want * (yv > 0)
        close(sums[0].sum(1), gm.sum(0), 2e-3)
        close(sums[1].sum(1), (gm * yv).sum(0), 2e-3)
    if k == 1:      # BatchNorm + ReLU on load: x holds z, channels with (1, 0) are activations already
        r = np.abs(rng.normal(size=Ci)) + 0.5
        sh = rng.normal(size=Ci) * 0.3
        zin = rng.normal(size=(N, H, W, Ci))
        yin = np.maximum(zin * r + sh, 0.0)
        nplan = ops.F32x3Plan(N, H, W, Ci, Ci, 1, 1, Co, Co)
        rt, st_ = dev(r), dev(sh)
        nplan.d.norm_rstd, nplan.d.norm_shift = rt.data_ptr(), st_.data_ptr()
        zn, zind = torch.empty(M, Co, device="cuda"), dev(zin)
        nplan.run(ops._p(zind), ops._p(wb), ops._p(zn))
        torch.cuda.synchronize()
        close(zn, S.conv2d_same(yin, w, 1).reshape(-1, Co))


def test_batchnorm_relu_applied_on_load_equals_the_materialised_activation():
    """The three kernel features behind InceptionV1Engine.zcat, each bit-identical to the materialised form:
    ds_conv_desc.norm_rstd / norm_shift (wide 1x1 kernel: x holds z, the loader applies relu(z*rstd + shift); channels
    with (1, 0) are activations already), mask_rstd / mask_shift (DS_EPI_BNSUMS epilogue rebuilds y from z), and
    ds_bn_bwd_apply with a row stride (a layer differentiated in place inside a wider buffer)."""
    ops = _ops()
    rng = np.random.RandomState(91)
    M, K, N = 20000, 192, 176                       # (enough row tiles for the wide kernel to be the one chosen)
    z_in = rng.normal(size=(M, K)).astype(np.float32)
    r = (np.abs(rng.normal(size=K)) + 0.5).astype(np.float32)
    sh = rng.normal(size=K).astype(np.float32) * 0.3
    r[:64], sh[:64] = 1.0, 0.0
    z_in[:, :64] = np.maximum(z_in[:, :64], 0)                          # the Branch_0 slice is an activation
    zt, rt, st = torch.from_numpy(z_in).cuda(), torch.from_numpy(r).cuda(), torch.from_numpy(sh).cuda()
    y_in = torch.empty(M, K, device="cuda")                               # the materialised form: ds_bn_apply_relu's output
    ops.bn_apply_relu(zt, M, K, rt, st, ops.make_segments([(0, K, y_in.data_ptr(), K)]))
    w = torch.from_numpy((rng.normal(size=(K, N)) * 0.1).astype(np.float32)).cuda()
    pivot = torch.from_numpy(rng.normal(size=N).astype(np.float32) * 0.1).cuda()
    outs = []
    for norm in (False, True):
        plan = ops.ConvPlan(M, 1, 1, K, K, 1, 1, 1, N, N, 0, 1, N, flags=ops.DS_EPI_STATS, pad_t=0, pad_l=0, OH=1, OW=1)
        if norm:
            assert ops.conv_norm_supported(plan)
            plan.d.norm_rstd, plan.d.norm_shift = rt.data_ptr(), st.data_ptr()
        z = torch.empty(M, N, device="cuda")
        stats = torch.zeros(2, N, plan.partials, device="cuda")
        plan.run(ops._p(zt if norm else y_in), ops._p(w), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot))
        torch.cuda.synchronize()
        outs.append((z, stats))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # dgrad with DS_EPI_BNSUMS: mask = z of the consumer (columns = K here) + per-column rstd / shift
    dz = torch.from_numpy(rng.normal(size=(M, N)).astype(np.float32)).cuda()
    res = []
    for norm in (False, True):
        g = ops.gemm_plan(M, N, K, N, K, N, transposed_w=True)
        P = g.enable_bnsums(K)
        assert P > 0
        if norm:
            g.d.mask_rstd, g.d.mask_shift = rt.data_ptr(), st.data_ptr()
        dx = torch.empty(M, K, device="cuda")
        sums = torch.zeros(2, K, P, device="cuda")
        g.run(ops._p(dz), ops._p(w), ops._p(dx), mask=ops._p(zt if norm else y_in), stats=ops._p(sums))
        torch.cuda.synchronize()
        res.append((dx, sums))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    # BatchNorm backward apply in place inside a wider buffer
    Cc, ld, off = 64, 160, 32
    zz = torch.from_numpy(rng.normal(size=(M, Cc)).astype(np.float32)).cuda()
    dy = torch.from_numpy(rng.normal(size=(M, Cc)).astype(np.float32)).cuda()
    mean, rstd, shift = (torch.from_numpy(rng.normal(size=Cc).astype(np.float32)).cuda() for _ in range(3))
    rstd = rstd.abs() + 0.5
    coef = torch.from_numpy(rng.normal(size=(2, Cc)).astype(np.float32) * 0.1).cuda()
    segs = ops.make_segments([(0, Cc, dy.data_ptr(), Cc)])
    dense = torch.empty(M, Cc, device="cuda")
    ops.bn_bwd_apply(zz, segs, M, Cc, mean, rstd, shift, coef, dense)
    wide = torch.full((M, ld), 7.0, device="cuda")
    wide[:, off:off + Cc] = zz
    view = wide[:, off:off + Cc]
    ops.bn_bwd_apply(view, segs, M, Cc, mean, rstd, shift, coef, view, ldz=ld)
    torch.cuda.synchronize()
    assert torch.equal(wide[:, off:off + Cc], dense)
    assert float((wide[:, :off] - 7.0).abs().max()) == 0.0 and float((wide[:, off + Cc:] - 7.0).abs().max()) == 0.0


def test_bn_backward_sums_from_mixed_sources_match_the_single_pass():
    """ds_bn_bwd_finalize_segs: a layer whose output gradient comes in three parts -- the sums of part 0 and part 2
    emitted by the producing dgrads (DS_EPI_BNSUMS form: sum g, sum g*y over y > 0, with different partial counts),
    part 1 reduced over its column range by ds_bn_bwd_reduce (ldz > C) -- gives the dbeta / mean(g) / mean(g*xhat) of
    the single-pass reduce over the whole layer, and of the oracle."""
    ops = _ops()
    rng = np.random.RandomState(77)
    M, Cc = 1500, 96
    parts = [(0, 40), (40, 72), (72, 96)]
    z = rng.normal(size=(M, Cc)) * 1.3 + rng.normal(size=Cc)
    beta = rng.normal(size=Cc) * 0.3
    mean, var = z.mean(0), z.var(0)
    rstd = 1.0 / np.sqrt(var + 1e-3)
    shift = beta - mean * rstd
    y = np.maximum(z * rstd + shift, 0.0)
    dy = rng.normal(size=(M, Cc))
    g = dy * (y > 0)
    xhat = (z - mean) * rstd
    zd, dyd = dev(z), dev(dy)
    mean_d, rstd_d, shift_d, beta_d = dev(mean), dev(rstd), dev(shift), dev(beta)
    # reference: the single-pass kernels
    P0 = ops.bn_bwd_partials(M, Cc)
    part = torch.empty(2 * Cc * P0, device="cuda")
    dbeta0, coef0 = torch.empty(Cc, device="cuda"), torch.empty(2, Cc, device="cuda")
    ops.bn_bwd_reduce(zd, ops.make_segments([(0, Cc, dyd.data_ptr(), Cc)]), M, Cc, mean_d, rstd_d, shift_d, part)
    ops.bn_bwd_finalize(part, P0, M, Cc, dbeta0, coef0)
    # mixed sources
    sg = ops.SumSegments()
    sg.nseg = 3
    keep = []
    for i, (c0, c1) in enumerate(parts):
        sg.c_begin[i], sg.c_end[i] = c0, c1
        n = c1 - c0
        if i == 1:
            scratch = torch.empty(2 * n * P0, device="cuda")
            ops.bn_bwd_reduce(C.c_void_p(zd.data_ptr() + 4 * c0), ops.make_segments([(0, n, dyd.data_ptr() + 4 * c0, Cc)]),
                              M, n, C.c_void_p(mean_d.data_ptr() + 4 * c0), C.c_void_p(rstd_d.data_ptr() + 4 * c0),
                              C.c_void_p(shift_d.data_ptr() + 4 * c0), C.c_void_p(scratch.data_ptr()), ldz=Cc)
            sg.P[i], sg.kind[i] = P0, 0
            sg.s[i], sg.q[i] = scratch.data_ptr(), scratch.data_ptr() + 4 * n * P0
            keep.append(scratch)
        else:
            # what a dgrad with DS_EPI_BNSUMS leaves behind: P row-block partials of sum g and sum g*y inside a wider
            # [2][Ctot][P] buffer (this part starts at column `off` of the producer's Ctot columns)
            P, off, ctot = (7, 8, 64) if i == 0 else (13, 0, n)
            rows = np.array_split(np.arange(M), P)
            buf = np.full((2, ctot, P), np.nan)
            for pi, r in enumerate(rows):
                buf[0, off:off + n, pi] = g[r][:, c0:c1].sum(0)
                buf[1, off:off + n, pi] = (g[r][:, c0:c1] * y[r][:, c0:c1]).sum(0)
            bd = dev(buf)
            sg.P[i], sg.kind[i] = P, 1
            sg.s[i] = bd.data_ptr() + 4 * off * P
            sg.q[i] = bd.data_ptr() + 4 * (ctot + off) * P
            keep.append(bd)
    dbeta, coef = torch.empty(Cc, device="cuda"), torch.empty(2, Cc, device="cuda")
    ops.bn_bwd_finalize_segs(sg, M, Cc, beta_d, dbeta, coef)
    torch.cuda.synchronize()
    close(dbeta, g.sum(0), 2e-5)
    close(coef[0], g.mean(0), 2e-5)
    close(coef[1], (g * xhat).mean(0), 1e-4)
    close(dbeta, dbeta0.cpu().numpy().astype(np.float64), 2e-6)
    close(coef[1], coef0[1].cpu().numpy().astype(np.float64), 1e-4)


def test_batch_norm_launches_of_three_layers_as_one_are_bit_identical():
    """ds_bn_finalize_multi / ds_bn_bwd_finalize_multi (round 6): the three convs that close an Inception block (Branch_1 /
    Branch_2 3x3, Branch_3 1x1, image_model/inception_v1.py:86-95) keep z and dy in column slices of one concat pair, so
    their BatchNorm finalizes go out as one launch each way and ONE ds_bn_bwd_apply covers the slices.  Layers with
    different partial counts, pivots, their own beta / dbeta / moving vectors; every output equals the per-layer launches'
    to the bit."""
    ops = _ops()
    rng = np.random.RandomState(17)
    M, b0, widths = 3000, 32, (96, 48, 32)
    Ct = b0 + sum(widths)
    Cb = Ct - b0
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    # ---- forward ----
    Ps = (24, 7, 130)
    layers = []
    for n, P in zip(widths, Ps):
        layers.append(dict(n=n, P=P, stats=dev(rng.normal(size=(2, n, P)) * 5 + np.array([0.0, 40.0])[:, None, None]),
                           beta=dev(rng.normal(size=n)), pivot=dev(rng.normal(size=n)), mm=dev(rng.normal(size=n)),
                           mv=dev(rng.uniform(0.5, 2, size=n))))
    outs = []
    for multi in (False, True):
        mean_cat, rs_cat = torch.zeros(Ct, device="cuda"), torch.zeros(2, Ct, device="cuda")
        jobs, off, keep = [], b0, []
        for l in layers:
            n = l["n"]
            mean = mean_cat[off:off + n]
            mean.copy_(l["pivot"])                       # the pivot aliases the mean, as in the engine
            mm, mv = l["mm"].clone(), l["mv"].clone()
            keep += [mm, mv]
            args = (l["stats"], l["P"], M, n, l["beta"], mean, mean, rs_cat[0, off:off + n], rs_cat[1, off:off + n], mm, mv)
            if multi:
                jobs.append(args)
            else:
                ops.bn_finalize(args[0], args[1], args[2], args[3], args[4], 1e-3, 0.9997, mean, args[7], args[8], mm, mv, pivot=mean)
            off += n
        if multi:
            ops.BnFinalizeJobs(jobs).run(1e-3, 0.9997)
        torch.cuda.synchronize()
        outs.append([mean_cat, rs_cat] + keep)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert float(outs[0][1][0, b0:].abs().min()) > 0
    # ---- backward: sums from a dgrad epilogue (kind 1) for two layers, reduced (kind 0) for one; then one apply ----
    mean_cat, rs_cat = outs[0][0], outs[0][1]
    z = dev(rng.normal(size=(M, Ct)))
    dy = dev(rng.normal(size=(M, Ct)))
    P1 = 11
    nxt = dev(rng.normal(size=(2, Ct, P1)))                     # [2][Ctot][P] of the next block's fused dgrad
    P0 = ops.bn_bwd_partials(M, widths[1])
    res = []
    for multi in (False, True):
        zc = z.clone()
        coef_cat = torch.zeros(2, Cb, device="cuda")
        dbetas = [torch.zeros(n, device="cuda") for n in widths]
        sgm = ops.SumSegments()
        sgm.nseg = 3
        off, keep = b0, []
        for i, l in enumerate(layers):
            n = l["n"]
            sg = ops.SumSegments()
            sg.nseg = 1
            sg.c_begin[0], sg.c_end[0] = 0, n
            if i == 1:
                scratch = torch.empty(2 * n * P0, device="cuda")
                ops.bn_bwd_reduce(C.c_void_p(zc.data_ptr() + 4 * off), ops.make_segments([(0, n, dy.data_ptr() + 4 * off, Ct)]), M, n,
                                  mean_cat[off:off + n], rs_cat[0, off:off + n], rs_cat[1, off:off + n], scratch, ldz=Ct)
                sg.P[0], sg.kind[0], sg.s[0], sg.q[0] = P0, 0, scratch.data_ptr(), scratch.data_ptr() + 4 * n * P0
                keep.append(scratch)
            else:
                sg.P[0], sg.kind[0] = P1, 1
                sg.s[0], sg.q[0] = nxt.data_ptr() + 4 * off * P1, nxt.data_ptr() + 4 * (Ct + off) * P1
            if multi:
                sgm.c_begin[i], sgm.c_end[i] = off - b0, off - b0 + n
                sgm.P[i], sgm.kind[i], sgm.s[i], sgm.q[i] = sg.P[0], sg.kind[0], sg.s[0], sg.q[0]
            else:
                coef = torch.empty(2, n, device="cuda")
                ops.bn_bwd_finalize_segs(sg, M, n, l["beta"], dbetas[i], coef)
                zs = zc[:, off:off + n]
                ops.bn_bwd_apply(zs, ops.make_segments([(0, n, dy.data_ptr() + 4 * off, Ct)]), M, n, mean_cat[off:off + n],
                                 rs_cat[0, off:off + n], rs_cat[1, off:off + n], coef, zs, ldz=Ct)
                coef_cat[:, off - b0:off - b0 + n] = coef
            off += n
        if multi:
            ops.bn_bwd_finalize_multi(sgm, M, Cb, [l["beta"] for l in layers], dbetas, coef_cat)
            zs = zc[:, b0:]
            ops.bn_bwd_apply(zs, ops.make_segments([(0, Cb, dy.data_ptr() + 4 * b0, Ct)]), M, Cb, mean_cat[b0:], rs_cat[0, b0:],
                             rs_cat[1, b0:], coef_cat, zs, ldz=Ct)
        torch.cuda.synchronize()
        res.append([zc, coef_cat] + dbetas)
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][0][:, :b0], z[:, :b0]) and not torch.equal(res[0][0][:, b0:], z[:, b0:])
    # a dbeta that is not wanted (the layer's beta is frozen): None in the list
    ops.bn_bwd_finalize_multi(sgm, M, Cb, [l["beta"] for l in layers], [None, res[1][3], None], torch.empty(2, Cb, device="cuda"))
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [(2, 14, 14, 96, 208, 1), (3, 7, 7, 160, 320, 3), (1, 28, 28, 64, 96, 3)])
def test_conv_and_batch_norm_passes_with_z_in_centred_bf16_storage(case):
    """ds_conv_desc.z_dtype = DS_DTYPE_BF16 + ds_bn_finalize_centered + ds_bn_apply_relu_z16 / ds_bn_bwd_reduce / ds_bn_bwd_apply_z16
    (round 6, the 16-bit labels): ds_conv_bf16 stores bf16(z - pivot) -- exactly the fp32 launch's z minus the pivot, rounded
    to nearest even -- with the SAME statistics; the finalize gives mean - pivot and beta - (mean - pivot) rstd beside the
    unchanged mean / rstd / shift; every BatchNorm pass over the bf16 tensor has the bits of the fp32-input pass over the same
    values.  And the point of the centring: xhat rebuilt from the stored z is within 2^-8 |xhat| + 2^-8 of the exact one however
    far the channel means are from zero (here a pivot that is NOT near the means is also covered: the error then follows
    |z - pivot|, which the assertion's bound on the stored value checks)."""
    ops = _ops()
    N, H, W, Ci, Co, k = case
    rng = np.random.RandomState(Ci + Co)
    M = N * H * W
    x = torch.relu(torch.from_numpy(rng.normal(size=(N, H, W, Ci)).astype(np.float32))).cuda()
    w = torch.from_numpy((rng.normal(size=(k, k, Ci, Co)) * 0.05).astype(np.float32)).cuda()
    # the pivot as the engine has it: the previous step's mean, i.e. close to this step's (here: the exact mean + 2 % of sigma)
    zref = torch.from_numpy(S.conv2d_same(_bf16_round(x.cpu().numpy().astype(np.float64)), _bf16_round(w.cpu().numpy().astype(np.float64)), 1).reshape(M, Co))
    pivot = (zref.mean(0) + 0.02 * zref.std(0)).float().cuda()
    outs = []
    for z16 in (False, True):
        pl = ops.LayerPlan(ops.DS_CONV_FWD, ops.DS_ARITH_BF16, ops.DS_PLAN_ACT16, N, H, W, Ci, Co, k, 1, Ci, Co, ops.DS_EPI_STATS)
        assert pl.family == ops.DS_FAM_BF16D
        pl.alloc_weights(x.device)
        pl.prepare(ops._p(w))
        z = torch.zeros(M, Co, device="cuda", dtype=torch.bfloat16 if z16 else torch.float32)
        pl.d.z_dtype = ops.DS_DTYPE_BF16 if z16 else ops.DS_DTYPE_F32
        stats = torch.zeros(2 * Co * pl.partials, device="cuda")
        pl.run(ops._p(x), ops._p(w), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot))
        torch.cuda.synchronize()
        outs.append((z, stats, pl.partials))
    z32, zc16, P = outs[0][0], outs[1][0], outs[0][2]
    close(z32, zref.numpy(), 5e-4)
    assert torch.equal(outs[0][1], outs[1][1])                                   # the statistics do not see the storage
    assert torch.equal(zc16, (z32 - pivot).to(torch.bfloat16))                   # bf16(z - pivot), RNE
    # finalize: unchanged outputs + the centred pair
    beta = torch.from_numpy(rng.normal(size=Co).astype(np.float32)).cuda()
    fin = []
    for centred in (False, True):
        mean, rstd, shift = pivot.clone(), torch.empty(Co, device="cuda"), torch.empty(Co, device="cuda")
        mm, mv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
        mc, sc = torch.zeros(Co, device="cuda"), torch.zeros(Co, device="cuda")
        if centred:
            ops.bn_finalize_centered(outs[0][1], P, M, Co, beta, 1e-3, 0.9997, mean, rstd, shift, mm, mv, mean, mc, sc)
        else:
            ops.bn_finalize(outs[0][1], P, M, Co, beta, 1e-3, 0.9997, mean, rstd, shift, mm, mv, pivot=mean)
        torch.cuda.synchronize()
        fin.append((mean, rstd, shift, mm, mv, mc, sc))
    for a, b in zip(fin[0][:5], fin[1][:5]):
        assert torch.equal(a, b)
    mean, rstd, shift, mc, sc = fin[1][0], fin[1][1], fin[1][2], fin[1][5], fin[1][6]
    assert float((mc.double() - (mean.double() - pivot.double())).abs().max()) <= 1e-6 * float(zref.std(0).max())
    close(sc, (beta.double() - mc.double() * rstd.double()).cpu().numpy(), 1e-6)
    # the passes over the bf16 tensor = the fp32-input passes over the same (centred, rounded) values
    zf = zc16.float()
    ya, yb = torch.zeros(M, Co, device="cuda"), torch.zeros(M, Co, device="cuda")
    ops.bn_apply_relu(zc16, M, Co, rstd, sc, ops.make_segments([(0, Co, ya.data_ptr(), Co)]))
    ops.bn_apply_relu(zf, M, Co, rstd, sc, ops.make_segments([(0, Co, yb.data_ptr(), Co)]))
    dy = torch.from_numpy(rng.normal(size=(M, Co)).astype(np.float32)).cuda()
    dy_segs = ops.make_segments([(0, Co, dy.data_ptr(), Co)])
    P0 = ops.bn_bwd_partials(M, Co)
    pa, pb = torch.zeros(2 * Co * P0, device="cuda"), torch.zeros(2 * Co * P0, device="cuda")
    ops.bn_bwd_reduce(zc16, dy_segs, M, Co, mc, rstd, sc, pa)
    ops.bn_bwd_reduce(zf, dy_segs, M, Co, mc, rstd, sc, pb)
    coef = torch.from_numpy(rng.normal(size=(2, Co)).astype(np.float32) * 0.01).cuda()
    da, db = torch.zeros(M, Co, device="cuda", dtype=torch.bfloat16), torch.zeros(M, Co, device="cuda", dtype=torch.bfloat16)
    ops.bn_bwd_apply(zc16, dy_segs, M, Co, mc, rstd, sc, coef, da)
    ops.bn_bwd_apply(zf.clone(), dy_segs, M, Co, mc, rstd, sc, coef, db)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb) and torch.equal(pa, pb) and torch.equal(da, db)
    assert float(ya.abs().max()) > 0 and float(da.float().abs().max()) > 0
    # no cancellation: xhat from the stored z against the exact one, although the means are many sigma from zero in some channels
    xh_exact = (z32.double() - mean.double()) * rstd.double()
    xh_stored = (zc16.double() - mc.double()) * rstd.double()
    err = (xh_stored - xh_exact).abs()
    assert float((err - 2.0 ** -8 * xh_exact.abs()).max()) <= 2.0 ** -8, float(err.max())


@pytest.mark.parametrize("out16", [False, True])
def test_batch_norm_backward_of_a_gradient_kept_in_two_tensors(out16):
    """ds_segments.ptr2 / ds_bn_sum_segments.P2 (round 6): an Inception block's input gradient is the sum of the fused 1x1 dgrad's
    output and Branch_3's pool gradient (image_model/inception_v1.py:83-96); where the dgrad cannot accumulate the two stay two
    tensors written on two streams.  The sums are linear, so each producer emits its addend's partials and the finalize adds the
    two sources; the apply pass adds the tensors as it reads.  dz (fp32 in place / bf16) is BIT-identical to the pass over the
    pre-added gradient; coefficients and dbeta equal the single-source finalize of the concatenated partials to rounding."""
    ops = _ops()
    rng = np.random.RandomState(5)
    M, Cc, c1 = 3000, 176, 64
    dev_ = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    z = dev_(rng.normal(size=(M, Cc)) * 2 + 0.3)
    mean, rstd, shift = dev_(rng.normal(size=Cc) * 0.1), dev_(rng.uniform(0.5, 2, size=Cc)), dev_(rng.normal(size=Cc) * 0.2)
    beta = dev_(rng.normal(size=Cc))
    ld = Cc + 16                                           # the two addends live in wider buffers (concat gradients)
    dya, dyb = dev_(rng.normal(size=(M, ld))), dev_(rng.normal(size=(M, ld)))
    dsum = dya + dyb
    # segment 0 (columns [0, c1)): two addends, sums from two kind-1 sources; segment 1: one tensor, one source
    P1, P2 = 7, 19
    srcs = [dev_(rng.normal(size=(2, Cc, P))) for P in (P1, P2)]
    cat = torch.cat([srcs[0][:, :c1], srcs[1][:, :c1]], dim=2).contiguous()       # [2][c1][P1 + P2]
    sg2, sg1 = ops.SumSegments(), ops.SumSegments()
    for sg in (sg2, sg1):
        sg.nseg = 2
        sg.c_begin[0], sg.c_end[0], sg.kind[0] = 0, c1, 1
        sg.c_begin[1], sg.c_end[1], sg.kind[1], sg.P[1] = c1, Cc, 1, P1
        sg.s[1], sg.q[1] = srcs[0].data_ptr() + 4 * c1 * P1, srcs[0].data_ptr() + 4 * (Cc + c1) * P1
    sg2.P[0], sg2.s[0], sg2.q[0] = P1, srcs[0].data_ptr(), srcs[0].data_ptr() + 4 * Cc * P1
    sg2.P2[0], sg2.s2[0], sg2.q2[0] = P2, srcs[1].data_ptr(), srcs[1].data_ptr() + 4 * Cc * P2
    sg1.P[0], sg1.s[0], sg1.q[0] = P1 + P2, cat.data_ptr(), cat.data_ptr() + 4 * c1 * (P1 + P2)
    res = []
    for two in (True, False):
        dbeta, coef = torch.zeros(Cc, device="cuda"), torch.zeros(2, Cc, device="cuda")
        ops.bn_bwd_finalize_segs(sg2 if two else sg1, M, Cc, beta, dbeta, coef)
        res.append((dbeta, coef))
    torch.cuda.synchronize()
    close(res[0][0], res[1][0].cpu().numpy().astype(np.float64), 1e-6)
    close(res[0][1], res[1][1].cpu().numpy().astype(np.float64), 1e-6)
    s_ref = (srcs[0][0, :c1].double().sum(1) + srcs[1][0, :c1].double().sum(1)).cpu().numpy()
    close(res[0][0][:c1], s_ref, 1e-6)
    coef = res[1][1]
    outs = []
    for two in (True, False):
        zc = z.clone()
        dz = torch.zeros(M, Cc, device="cuda", dtype=torch.bfloat16) if out16 else zc
        src = dya if two else dsum
        segs = ops.make_segments([(0, c1, src.data_ptr(), ld), (c1, Cc, dsum.data_ptr() + 4 * c1, ld)])
        if two:
            segs.ptr2[0] = dyb.data_ptr()
        ops.bn_bwd_apply(zc, segs, M, Cc, mean, rstd, shift, coef, dz)
        torch.cuda.synchronize()
        outs.append(dz.clone())
    assert torch.equal(outs[0], outs[1])
    assert not torch.equal(outs[0].float(), z)
    # the reduce pass refuses a gradient in two tensors (its sums come from the producers)
    segs = ops.make_segments([(0, Cc, dya.data_ptr(), ld)])
    segs.ptr2[0] = dyb.data_ptr()
    with pytest.raises(RuntimeError):
        ops.bn_bwd_reduce(z, segs, M, Cc, mean, rstd, shift, torch.empty(2 * Cc * ops.bn_bwd_partials(M, Cc), device="cuda"))


@pytest.mark.parametrize("case", [(3000, 176, 24), (25088, 624, 196), (70, 16, 3), (200704, 64, 1568)])
def test_batch_norm_finalize_and_apply_as_one_launch_are_bit_identical(case):
    """ds_bn_finalize_apply_relu / ds_bn_bwd_finalize_apply (round 6): the finalize (one workgroup per channel) and the apply pass
    that reads its result as ONE launch -- the apply workgroups wait on a device-side ticket instead of a dependent-launch
    boundary (slim.batch_norm + relu of every conv, slim/nets/inception_utils.py:48-70).  Every output -- statistics, moving
    averages, activations in fp32 and bf16 segments, dbeta, coefficients, dz in fp32 (in place) and bf16 -- equals the two
    launches' to the bit; called three times in a row (the launch leaves its ticket words zero)."""
    ops = _ops()
    M, Cc, P = case
    rng = np.random.RandomState(M % 1000 + Cc)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    z = dev(rng.normal(size=(M, Cc)) * 2 + 0.3)
    stats = dev(rng.normal(size=(2, Cc, P)) * 5 + np.array([0.0, 40.0])[:, None, None])
    beta, pivot = dev(rng.normal(size=Cc)), dev(rng.normal(size=Cc))
    c1 = (Cc // 8) * 4
    ticket = torch.zeros(4, dtype=torch.int32, device="cuda")

    def forward(merged):
        mean, rstd, shift = pivot.clone(), torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
        mm, mv = beta.clone(), beta.abs() + 1
        ya, yb = torch.zeros(M, c1, device="cuda"), torch.zeros(M, Cc, device="cuda", dtype=torch.bfloat16)
        segs = ops.make_segments([(0, c1, ya.data_ptr(), c1), (c1, Cc, yb.data_ptr() + 2 * c1, Cc, ops.DS_DTYPE_BF16, None)])
        if merged:
            ops.bn_finalize_apply_relu(stats, P, M, Cc, beta, 1e-3, 0.9997, mean, rstd, shift, mm, mv, mean, z, M, segs, ticket[0:2])
        else:
            ops.bn_finalize(stats, P, M, Cc, beta, 1e-3, 0.9997, mean, rstd, shift, mm, mv, pivot=mean)
            ops.bn_apply_relu(z, M, Cc, rstd, shift, segs)
        torch.cuda.synchronize()
        return [mean, rstd, shift, mm, mv, ya, yb]

    want = forward(False)
    for _ in range(3):
        got = forward(True)
        assert all(torch.equal(a, b) for a, b in zip(want, got))
        assert int(ticket.abs().sum()) == 0
    assert float(want[5].abs().max()) > 0 and float(want[6][:, c1:].float().abs().max()) > 0
    mean, rstd, shift = want[0], want[1], want[2]
    # ---- backward: one segment from reduce partials (kind 0), one in the DS_EPI_BNSUMS form (kind 1); dz fp32 in place / bf16
    dy = dev(rng.normal(size=(M, Cc)))
    dy_segs = ops.make_segments([(0, Cc, dy.data_ptr(), Cc)])
    P0 = ops.bn_bwd_partials(M, c1)
    scratch = torch.empty(2 * c1 * P0, device="cuda")
    ops.bn_bwd_reduce(C.c_void_p(z.data_ptr()), ops.make_segments([(0, c1, dy.data_ptr(), Cc)]), M, c1, mean, rstd, shift, scratch, ldz=Cc)
    P1 = 9
    nxt = dev(rng.normal(size=(2, Cc, P1)))
    sg = ops.SumSegments()
    sg.nseg = 2
    sg.c_begin[0], sg.c_end[0], sg.P[0], sg.kind[0] = 0, c1, P0, 0
    sg.s[0], sg.q[0] = scratch.data_ptr(), scratch.data_ptr() + 4 * c1 * P0
    sg.c_begin[1], sg.c_end[1], sg.P[1], sg.kind[1] = c1, Cc, P1, 1
    sg.s[1], sg.q[1] = nxt.data_ptr() + 4 * c1 * P1, nxt.data_ptr() + 4 * (Cc + c1) * P1

    def backward(merged, out16):
        zc = z.clone()
        dz = torch.zeros(M, Cc, device="cuda", dtype=torch.bfloat16) if out16 else zc
        dbeta, coef = torch.zeros(Cc, device="cuda"), torch.zeros(2, Cc, device="cuda")
        if merged:
            ops.bn_bwd_finalize_apply(sg, M, Cc, beta, dbeta, coef, zc, dy_segs, mean, rstd, shift, dz, ticket[2:4])
        else:
            ops.bn_bwd_finalize_segs(sg, M, Cc, beta, dbeta, coef)
            ops.bn_bwd_apply(zc, dy_segs, M, Cc, mean, rstd, shift, coef, dz)
        torch.cuda.synchronize()
        return [dbeta, coef, dz, zc]

    for out16 in (False, True):
        want = backward(False, out16)
        for _ in range(3):
            got = backward(True, out16)
            assert all(torch.equal(a, b) for a, b in zip(want, got)), out16
            assert int(ticket.abs().sum()) == 0
        assert not torch.equal(want[2].float(), z)


@pytest.mark.parametrize("case", [(5, 14, 14, 480, "f32"), (3, 28, 28, 256, "bf16"), (9, 7, 7, 832, "f32"), (2, 9, 5, 12, "bf16"),
                                  (300, 7, 7, 64, "bf16"), (6, 14, 14, 528, "bf16"), (2, 5, 7, 1024, "f32"), (3, 4, 4, 68, "bf16")])
def test_max_pool_gradient_that_also_emits_the_batch_norm_sums(case):
    """ds_maxpool3_bwd_sums (round 6): MaxPoolGrad of Branch_3's 3x3 / 1 pool (inception_v1.py:94 ... :246) added LAST onto the
    block-input gradient also leaves the previous block's BatchNorm-backward sums -- sum g and sum g*y with g = dx (y > 0),
    the DS_EPI_BNSUMS form -- so that block's ds_bn_bwd_reduce passes go.  dx is bit-identical to ds_maxpool_bwd (accumulating
    and not); the partials add up to the fp64 sums; fp32 and bf16 activations; widths the launch splits into channel chunks (528: 6 x 22
    quads, 832: 13 x 16, 480: 2 x 60) and widths that leave threads idle (12, 68)."""
    ops = _ops()
    N, H, W, Cc, dt = case
    rng = np.random.RandomState(N + Cc)
    x = torch.from_numpy(rng.normal(size=(N, H, W, Cc)).astype(np.float32)).cuda()
    pooled, am = torch.empty_like(x), torch.empty(N, H, W, Cc, dtype=torch.uint8, device="cuda")
    ops.maxpool_fwd(x, pooled, am, N, H, W, Cc, 3, 1, "SAME")
    dy = torch.from_numpy(rng.normal(size=(N, H, W, Cc)).astype(np.float32)).cuda()
    base = torch.from_numpy(rng.normal(size=(N, H, W, Cc)).astype(np.float32)).cuda()
    y = torch.relu(torch.from_numpy(rng.normal(size=(N, H, W, Cc)).astype(np.float32)).cuda())
    if dt == "bf16":
        y = y.to(torch.bfloat16)
    P = ops.maxpool3_bwd_sums_partials(N, W, Cc)
    assert 1 <= P <= N * W
    for acc in (True, False):
        want, got = base.clone(), base.clone()
        ops.maxpool_bwd(dy, am, want, acc, N, H, W, Cc, 3, 1, "SAME")
        part = torch.full((2, Cc, P), float("nan"), device="cuda")
        ops.maxpool3_bwd_sums(dy, am, got, acc, y, N, H, W, Cc, part)
        torch.cuda.synchronize()
        assert torch.equal(want, got)
        g = (want.double() * (y.double() > 0)).reshape(-1, Cc)
        yy = y.double().reshape(-1, Cc)
        s_ref, q_ref = g.sum(0).cpu().numpy(), (g * yy).sum(0).cpu().numpy()
        scale_s = float(g.abs().sum(0).max()) + 1e-30
        scale_q = float((g * yy).abs().sum(0).max()) + 1e-30
        s_got, q_got = part[0].double().sum(1).cpu().numpy(), part[1].double().sum(1).cpu().numpy()
        assert np.isfinite(s_got).all() and np.isfinite(q_got).all()
        assert np.abs(s_got - s_ref).max() <= 2e-6 * scale_s, np.abs(s_got - s_ref).max() / scale_s
        assert np.abs(q_got - q_ref).max() <= 2e-6 * scale_q, np.abs(q_got - q_ref).max() / scale_q


@pytest.mark.parametrize("case", [(4, 14, 14, 480), (2, 28, 28, 192), (9, 7, 7, 832), (3, 6, 5, 12)])
def test_max_pool_gradient_reading_bf16_storage_and_a_dgrad_writing_it(case):
    """ds_maxpool3_bwd_dy16 + ds_conv_desc.z_dtype on a Conv2DBackpropInput launch (round 6, 16-bit labels, off in the engine:
    no time gain): Branch_3's 1x1 dgrad writes its output rounded to bf16 and the 3x3 / 1 MaxPoolGrad reads that storage.  The
    pool gradient (plain and accumulating, and with the BatchNorm sums) has the bits of ds_maxpool_bwd / ds_maxpool3_bwd_sums
    on the same values in fp32; the dgrad's bf16 output is its fp32 output rounded to nearest even."""
    ops = _ops()
    N, H, W, Cc = case
    rng = np.random.RandomState(N * 7 + Cc)
    x = torch.from_numpy(rng.normal(size=(N, H, W, Cc)).astype(np.float32)).cuda()
    pooled, am = torch.empty_like(x), torch.empty(N, H, W, Cc, dtype=torch.uint8, device="cuda")
    ops.maxpool_fwd(x, pooled, am, N, H, W, Cc, 3, 1, "SAME")
    dy16 = torch.from_numpy(rng.normal(size=(N, H, W, Cc)).astype(np.float32)).cuda().to(torch.bfloat16)
    dy32 = dy16.float()
    base = torch.from_numpy(rng.normal(size=(N, H, W, Cc)).astype(np.float32)).cuda()
    y = torch.relu(torch.from_numpy(rng.normal(size=(N, H, W, Cc)).astype(np.float32)).cuda()).to(torch.bfloat16)
    P = ops.maxpool3_bwd_sums_partials(N, W, Cc)
    for acc in (True, False):
        a, b = base.clone(), base.clone()
        ops.maxpool_bwd(dy32, am, a, acc, N, H, W, Cc, 3, 1, "SAME")
        ops.maxpool_bwd(dy16, am, b, acc, N, H, W, Cc, 3, 1, "SAME")
        pa, pb = torch.zeros(2, Cc, P, device="cuda"), torch.zeros(2, Cc, P, device="cuda")
        c, d = base.clone(), base.clone()
        ops.maxpool3_bwd_sums(dy32, am, c, acc, y, N, H, W, Cc, pa)
        ops.maxpool3_bwd_sums(dy16, am, d, acc, y, N, H, W, Cc, pb)
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(c, d) and torch.equal(a, c) and torch.equal(pa, pb)
    if Cc % 16 == 0:
        Ci = 64          # a 1x1 dgrad [M, Ci] -> [M, Cc] on the register-direct bf16 kernel, fp32 and bf16 output
        M = N * H * W
        w = torch.from_numpy((rng.normal(size=(1, 1, Cc, Ci)) * 0.05).astype(np.float32)).cuda()
        dz = torch.from_numpy(rng.normal(size=(M, Ci)).astype(np.float32)).cuda()
        outs = []
        for o16 in (False, True):
            pl = ops.LayerPlan(ops.DS_CONV_DGRAD, ops.DS_ARITH_BF16, ops.DS_PLAN_ACT16, N, H, W, Cc, Ci, 1, 1, Ci, Cc, 0)
            assert pl.family == ops.DS_FAM_BF16D
            pl.alloc_weights(x.device)
            pl.prepare(ops._p(w))
            out = torch.zeros(M, Cc, device="cuda", dtype=torch.bfloat16 if o16 else torch.float32)
            pl.d.z_dtype = ops.DS_DTYPE_BF16 if o16 else ops.DS_DTYPE_F32
            pl.run(ops._p(dz), ops._p(w), ops._p(out))
            torch.cuda.synchronize()
            outs.append(out)
        assert torch.equal(outs[1], outs[0].to(torch.bfloat16)) and float(outs[0].abs().max()) > 0


def _fp8_round(a, fmax, mant, emin):
    """saturating round-to-nearest-even to an OCP fp8 format (e4m3fn: 448, 3, -6; e5m2: 57344, 2, -14), as float64"""
    a = np.asarray(a, np.float64)
    m = np.minimum(np.abs(a), fmax)
    e = np.maximum(np.frexp(m)[1] - 1, emin)
    step = np.ldexp(1.0, e - mant)
    return np.sign(a) * np.rint(m / step) * step


def _pow2_scale(amax, fmax):
    if not amax > 0:
        return 1.0
    r = np.float32(fmax) / np.float32(amax)
    return float(2.0 ** (int(np.frexp(r)[1]) - 1))


def test_absmax_and_fp8_formats_on_this_device():
    """ds_absmax is exact; and what v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 + the fp8 MFMA do on THIS device is the OCP
    e4m3fn / e5m2 arithmetic the oracle emulates (gfx942 would be FNUZ: a factor 2 here): a 32 x 16 x 32 product of
    values that are exactly representable comes out exact, values between two codes round to nearest even (ties
    included), subnormals keep their absolute step."""
    ops = _ops()
    rng = np.random.RandomState(3)
    x = rng.normal(size=100003).astype(np.float32) * 3
    x[777] = -41.5
    out = torch.zeros(ops.AMAX_FLOATS, device="cuda")
    xd = dev(x)
    ops.absmax(xd, x.size, out)
    torch.cuda.synchronize()
    assert ops.amax_value(out) == 41.5
    ops.absmax(xd.to(torch.bfloat16), x.size - 1, out)             # bf16 storage (an even count)
    torch.cuda.synchronize()
    assert ops.amax_value(out) == 41.5
    # one tile: M = 32, K = 16, N = 32, identity-like weights pick single products
    for a_format, fmax, mant, emin in ((ops.DS_FP8_E4M3, 448.0, 3, -6), (ops.DS_FP8_E5M2, 57344.0, 2, -14)):
        M, K, N = 32, 16, 32
        # column 0 pins amax = fmax (s_a = 1); 1.0625 / 1.1875 are ties in e4m3, 1.125 is one in e5m2; subnormals
        vals = [fmax, 1.0, 1.125, 1.0625, 1.1875, 0.3, -2.7, 300.0, 2.0 ** emin, 2.0 ** (emin - 2), 1e-4, 17.0, -0.5, 3.3,
                100.0, 0.0]
        xs = np.tile(np.array(vals, np.float32)[None, :], (M, 1))
        w = np.zeros((1, 1, K, N), np.float32)
        for k in range(K):
            w[0, 0, k, k] = 1.0                            # column k = operand k
        w[0, 0, 0, 16] = 448.0                             # pins the weight amax -> s_w = 1
        plan = ops.Fp8Plan(M, 1, 1, K, K, 1, 1, N, N, a_format=a_format, pad_t=0, pad_l=0, OH=1, OW=1)
        wq = torch.empty(ops.weights_fp8_bytes(K, N, 1, False), dtype=torch.uint8, device="cuda")
        ws = torch.zeros(ops.WSCALE_FLOATS, device="cuda")
        ops.weights_to_fp8(ops._p(dev(w)), wq, ws, K, N, 1, False)
        xd = dev(xs)
        amax = torch.zeros(ops.AMAX_FLOATS, device="cuda")
        ops.absmax(xd, xs.size, amax)
        z = torch.empty(M, N, device="cuda")
        plan.run(ops._p(xd), ops._p(wq), ops._p(z), x_amax=ops._p(amax), wscale=ops._p(ws))
        torch.cuda.synchronize()
        assert float(ws[1].item()) == 1.0 and ops.amax_value(amax) == fmax and float(ws[0].item()) == 448.0
        want = _fp8_round(xs, fmax, mant, emin)
        got = z.cpu().numpy().astype(np.float64)[:, :K]
        np.testing.assert_array_equal(got, want, err_msg="format %d" % a_format)


def test_storage_and_amax_side_outputs_of_the_streaming_kernels():
    """What the bf16 / fp8 configurations add to the HBM-bound kernels: ds_bn_apply_relu writes a destination segment
    as bf16 (round to nearest even) and raises the segment's max|.| record; ds_bn_bwd_apply raises max|dz|; the forward
    pools read and write bf16 storage with the same winners; the pool fused with BatchNorm + ReLU writes bf16 and raises
    its record.  Values against NumPy, records exact (a maximum is order independent)."""
    ops = _ops()
    rng = np.random.RandomState(123)
    M, Cc = 777, 48
    z = rng.normal(size=(M, Cc)) * 2
    rstd = np.abs(rng.normal(size=Cc)) + 0.5
    shift = rng.normal(size=Cc) * 0.3
    y = np.maximum(z * rstd + shift, 0.0)
    zd, rd, sd = dev(z), dev(rstd), dev(shift)
    out16 = torch.zeros(M, 32, dtype=torch.bfloat16, device="cuda")
    out32 = torch.zeros(M, 16, device="cuda")
    am16, am32 = torch.zeros(ops.AMAX_FLOATS, device="cuda"), torch.zeros(ops.AMAX_FLOATS, device="cuda")
    segs = ops.make_segments([(0, 32, out16.data_ptr(), 32, ops.DS_DTYPE_BF16, ops._p(am16)),
                              (32, 48, out32.data_ptr(), 16, ops.DS_DTYPE_F32, ops._p(am32))])
    ops.bn_apply_relu(zd, M, Cc, rd, sd, segs)
    torch.cuda.synchronize()
    close(out32, y[:, 32:], 1e-6)
    close(out16.float(), y[:, :32], 4e-3)                    # bf16: 8 significant bits
    # every stored value is the bf16 rounding of an fp32 value within rounding of the oracle's
    assert float((out16.float() - dev(y[:, :32])).abs().max()) <= 2.0 ** -8 * float(y[:, :32].max())
    assert ops.amax_value(am32) == float(out32.max())        # the record holds the fp32 maximum, before any rounding
    assert abs(ops.amax_value(am16) - float(y[:, :32].max())) <= 1e-6 * float(y[:, :32].max())
    # BatchNorm backward apply with max|dz|
    dy = rng.normal(size=(M, Cc))
    mean = z.mean(0)
    coef = rng.normal(size=(2, Cc)) * 0.01
    dyd, md, cd = dev(dy), dev(mean), dev(coef)
    dz = torch.empty(M, Cc, device="cuda")
    amz = torch.zeros(ops.AMAX_FLOATS, device="cuda")
    ops.bn_bwd_apply(zd, ops.make_segments([(0, Cc, dyd.data_ptr(), Cc)]), M, Cc, md, rd, sd, cd, dz, amax=amz)
    torch.cuda.synchronize()
    g = dy * (y > 0)
    close(dz, rstd * (g - coef[0] - (z - mean) * rstd * coef[1]), 1e-5)
    assert ops.amax_value(amz) == float(dz.abs().max())
    # pools on bf16 storage: same values and winners as on the widened input
    N, H, W, C_ = 3, 9, 9, 16
    x16 = dev(np.maximum(rng.normal(size=(N, H, W, C_)), 0)).to(torch.bfloat16)
    for k, stride in ((3, 1), (3, 2), (2, 2)):
        OH = -(-H // stride) if k == 3 else H // 2
        outs = []
        for x in (x16, x16.float()):
            yv = torch.zeros(N, OH, OH, C_, dtype=x.dtype, device="cuda")
            am = torch.zeros(N, OH, OH, C_, dtype=torch.uint8, device="cuda")
            ops.maxpool_fwd(x, yv, am, N, H, W, C_, k, stride, "SAME" if k == 3 else "VALID")
            outs.append((yv.float(), am))
        torch.cuda.synchronize()
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (k, stride)
    zc = dev(rng.normal(size=(N, H, W, C_)))
    r2, s2 = dev(np.abs(rng.normal(size=C_)) + 0.5), dev(rng.normal(size=C_) * 0.2)
    res = []
    for dt in (torch.bfloat16, torch.float32):
        yv = torch.zeros(N, 5, 5, C_, dtype=dt, device="cuda")
        am = torch.zeros(N, 5, 5, C_, dtype=torch.uint8, device="cuda")
        rec = torch.zeros(ops.AMAX_FLOATS, device="cuda")
        ops.maxpool_bn_relu_fwd(zc, r2, s2, yv, am, N, H, W, C_, 3, 2, amax=rec)
        torch.cuda.synchronize()
        res.append((yv, am, ops.amax_value(rec)))
    assert torch.equal(res[0][0], res[1][0].to(torch.bfloat16)) and torch.equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] == float(res[1][0].max())


FP8_CASES = [
    # (N, H, W, Cin, Cout, k)
    (2, 9, 9, 16, 32, 1),
    (3, 14, 14, 24, 64, 3),
    (2, 28, 28, 96, 128, 3),
    (4, 7, 7, 832, 624, 1),
    (2, 13, 11, 48, 176, 3),
    (1, 8, 8, 8, 200, 1),
]


@pytest.mark.parametrize("case", FP8_CASES)
def test_fp8_conv_forward_and_dgrad_match_quantising_oracle(case):
    """ds_conv_fp8 against the fp64 convolution of the SAME quantised operands (per-tensor power-of-two scale from
    max|.|, saturating round-to-nearest-even to e4m3 / e5m2): products of fp8 values are exact in fp32, so only the
    accumulation separates the two -- measured 3.5e-5 of the largest output on MI355X (the fp8 MFMA sums its 16 products
    with ~15 bits of alignment before the fp32 accumulate), gate 1e-4; forward with BatchNorm statistics, the
    input gradient through the flipped / transposed filter with e5m2 gradients.  Against the UNQUANTISED convolution
    the error is the formats' own: ~4 % (e4m3 x e4m3) and ~8 % (e5m2 x e4m3) relative L2, printed."""
    ops = _ops()
    N, H, W, Ci, Co, k = case
    rng = np.random.RandomState(9)
    x = (np.maximum(rng.normal(size=(N, H, W, Ci)), 0) * 1.7).astype(np.float32)         # post-ReLU like
    w = (rng.normal(size=(k, k, Ci, Co)) * 0.05).astype(np.float32)
    sx, sw = _pow2_scale(np.abs(x).max(), 448.0), _pow2_scale(np.abs(w).max(), 448.0)
    xq, wq_ = _fp8_round(x.astype(np.float64) * sx, 448.0, 3, -6), _fp8_round(w.astype(np.float64) * sw, 448.0, 3, -6)
    ref = S.conv2d_same(xq, wq_, 1) / (sx * sw)
    exact = S.conv2d_same(x.astype(np.float64), w.astype(np.float64), 1)
    xd, wd = dev(x), dev(w)
    M = N * H * W
    plan = ops.Fp8Plan(N, H, W, Ci, Ci, k, 1, Co, Co, flags=ops.DS_EPI_STATS, a_format=ops.DS_FP8_E4M3)
    wq = torch.empty(ops.weights_fp8_bytes(Ci, Co, k * k, False), dtype=torch.uint8, device="cuda")
    ws = torch.zeros(ops.WSCALE_FLOATS, device="cuda")
    ops.weights_to_fp8(ops._p(wd), wq, ws, Ci, Co, k * k, False)
    amax = torch.zeros(ops.AMAX_FLOATS, device="cuda")
    ops.absmax(xd, x.size, amax)
    z = torch.full((M, Co), float("nan"), device="cuda")
    stats = torch.zeros(2, Co, plan.partials, device="cuda")
    pivot = dev(rng.normal(size=Co) * 0.1)
    plan.run(ops._p(xd), ops._p(wq), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot), x_amax=ops._p(amax), wscale=ops._p(ws))
    torch.cuda.synchronize()
    assert float(ws[1].item()) == sw
    zz = ref.reshape(M, Co)
    close(z, zz, 1e-4)
    pv = pivot.cpu().numpy().astype(np.float64)
    close(stats[0].sum(1), (zz - pv).sum(0), 2e-3)
    close(stats[1].sum(1), ((zz - pv) ** 2).sum(0), 2e-3)
    e_fwd = np.linalg.norm(z.cpu().numpy().reshape(exact.shape) - exact) / np.linalg.norm(exact)
    # dgrad: e5m2 gradients x e4m3 flipped / transposed filter
    if Co % 8 == 0:
        dy = (rng.normal(size=ref.shape) * 3e-4).astype(np.float32)
        sd = _pow2_scale(np.abs(dy).max(), 57344.0)
        dq = _fp8_round(dy.astype(np.float64) * sd, 57344.0, 2, -14)
        dref = S.conv2d_same_bwd_input(dq, wq_, (N, H, W, Ci), 1) / (sd * sw)
        dexact = S.conv2d_same_bwd_input(dy.astype(np.float64), w.astype(np.float64), (N, H, W, Ci), 1)
        g = ops.Fp8Plan(N, H, W, Co, Co, k, 1, Ci, Ci, a_format=ops.DS_FP8_E5M2)
        wqd = torch.empty(ops.weights_fp8_bytes(Ci, Co, k * k, True), dtype=torch.uint8, device="cuda")
        wsd = torch.zeros(ops.WSCALE_FLOATS, device="cuda")
        ops.weights_to_fp8(ops._p(wd), wqd, wsd, Ci, Co, k * k, True)
        dyd = dev(dy)
        ops.absmax(dyd, dy.size, amax)
        dx = torch.full((M, Ci), float("nan"), device="cuda")
        g.run(ops._p(dyd), ops._p(wqd), ops._p(dx), x_amax=ops._p(amax), wscale=ops._p(wsd))
        torch.cuda.synchronize()
        close(dx, dref.reshape(M, Ci), 1e-4)

        def run(flags, ldmask, mask_dtype, out, mask):
            g2 = ops.Fp8Plan(N, H, W, Co, Co, k, 1, Ci, Ci, flags=flags, a_format=ops.DS_FP8_E5M2)
            g2.d.ldmask, g2.d.mask_dtype = ldmask, mask_dtype
            P = (M + 127) // 128
            sums = torch.full((2, Ci, P), float("nan"), device="cuda")
            g2.run(ops._p(dyd), ops._p(wqd), ops._p(out), stats=ops._p(sums), x_amax=ops._p(amax), wscale=ops._p(wsd), mask=ops._p(mask))
            return sums, P
        _check_accum_bnsums_epilogue(run, dref.reshape(M, Ci), M, Ci, rng, 1e-4)
        e_bwd = np.linalg.norm(dx.cpu().numpy().reshape(dexact.shape) - dexact) / np.linalg.norm(dexact)
        print("fp8 conv %s: relative L2 against the unquantised convolution: forward %.3f, dgrad %.3f" % (case, e_fwd, e_bwd))
        assert e_fwd <= 0.08 and e_bwd <= 0.15


def _bf16_round(a):
    """round-to-nearest-even to bfloat16, returned as float64 (what v_cvt_pk_bf16_f32 does to the operands)"""
    u = np.asarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


BF16_CASES = [
    # (N, H, W, Cin, Cout, k, stride)
    (2, 9, 9, 16, 32, 1, 1),
    (3, 14, 14, 24, 64, 3, 1),       # Cin not a multiple of the 32-wide K tile
    (2, 28, 28, 96, 128, 3, 1),
    (4, 7, 7, 832, 624, 1, 1),       # several 128-wide column tiles, ragged last one
    (2, 13, 11, 48, 176, 3, 1),
    (1, 8, 8, 8, 200, 1, 1),
    (2, 10, 10, 20, 12, 3, 2),
    (2, 150, 150, 16, 96, 3, 1),     # 3-wide column tile
    (2, 370, 370, 8, 64, 3, 1),      # M = 273800: the persistent launch path (one workgroup walks several row tiles)
]


@pytest.mark.parametrize("case", BF16_CASES)
def test_conv_bf16_forward_dgrad_match_oracle(case):
    """DS_DTYPE_BF16 (v_mfma_f32_32x32x16_bf16, fp32 accumulate): forward with BatchNorm statistics and the dgrad
    of the same geometry.  Two gates: (a) against the fp64 oracle evaluated on bf16-ROUNDED operands the result is
    fp32-accumulation exact (2e-4 of max|ref|) -- layout, transposition, tails, statistics all right; (b) against
    the oracle on the unrounded operands the documented bf16 tolerance, 1e-2 of max|ref|."""
    ops = _ops()
    N, H, W, Ci, Co, k, s = case
    rng = np.random.RandomState(3)
    x = rng.normal(size=(N, H, W, Ci))
    w = rng.normal(size=(k, k, Ci, Co)) * 0.1
    ref_exact = S.conv2d_same(x, w, s)
    ref_round = S.conv2d_same(_bf16_round(x), _bf16_round(w), s)
    xd, wd = dev(x), dev(w)
    plan = ops.ConvPlan(N, H, W, Ci, Ci, k, k, s, Co, Co, Ci * Co, 1, Co, flags=ops.DS_EPI_STATS, dtype=ops.DS_DTYPE_BF16)
    M = plan.M
    z = torch.empty(M, Co, device="cuda")
    stats = torch.zeros(2, Co, plan.partials, device="cuda")
    plan.run(ops._p(xd), ops._p(wd), ops._p(z), stats=ops._p(stats))
    torch.cuda.synchronize()
    close(z, ref_round.reshape(M, Co), 2e-4)
    close(z, ref_exact.reshape(M, Co), 1e-2)
    zz = ref_round.reshape(M, Co)
    close(stats[0].sum(1), zz.sum(0), 2e-3)
    close(stats[1].sum(1), (zz ** 2).sum(0), 2e-3)
    if s == 1:
        dy = rng.normal(size=ref_exact.shape)
        g = ops.ConvPlan(N, H, W, Co, Co, k, k, 1, Ci, Ci, Ci * Co, Co, 1, flip=1, dtype=ops.DS_DTYPE_BF16)
        dx = torch.empty(g.M, Ci, device="cuda")
        dyd = dev(dy)
        g.run(ops._p(dyd), ops._p(wd), ops._p(dx))
        torch.cuda.synchronize()
        close(dx, S.conv2d_same_bwd_input(_bf16_round(dy), _bf16_round(w), (N, H, W, Ci), 1).reshape(-1, Ci), 2e-4)
        close(dx, S.conv2d_same_bwd_input(dy, w, (N, H, W, Ci), 1).reshape(-1, Ci), 1e-2)


def _check_accum_bnsums_epilogue(run, dx_ref, M, C_, rng, tol):
    """The dgrad epilogue of the register-direct bf16 / fp8 kernels (the wide fp32 kernel's): dx += result next to
    DS_EPI_BNSUMS -- column sums of g = dx_final (y > 0) and g y -- with y in fp32 and in bf16 storage.
    run(flags, ldmask, mask_dtype, dx, mask, sums) launches; dx_ref is the plain dgrad result (fp64)."""
    ops = _ops()
    prev = rng.normal(size=(M, C_)) * np.abs(dx_ref).max()
    yv = np.maximum(rng.normal(size=(M, C_)), 0.0) * (rng.uniform(size=(M, C_)) < 0.7)
    full = prev + dx_ref
    for dt in (ops.DS_DTYPE_F32, ops.DS_DTYPE_BF16):
        ld = C_ + 8
        ypad = np.pad(yv, ((0, 0), (0, 8)), constant_values=3.0)
        yd = dev(ypad) if dt == ops.DS_DTYPE_F32 else dev(ypad, torch.bfloat16)
        y_seen = yd.float().cpu().numpy().astype(np.float64)[:, :C_]          # (bf16 storage rounds y)
        dx = dev(prev)
        sums, P = run(ops.DS_EPI_ACCUM | ops.DS_EPI_BNSUMS, ld, dt, dx, yd)
        torch.cuda.synchronize()
        close(dx, full, tol)
        gm = dx.cpu().numpy().astype(np.float64) * (y_seen > 0)               # sums of what the kernel stored
        close(sums[0].sum(1), gm.sum(0), 2e-3)
        close(sums[1].sum(1), (gm * y_seen).sum(0), 2e-3)


BF16D_CASES = [
    # (N, H, W, Cin, Cout, k, stride)
    (2, 9, 9, 16, 32, 1, 1),
    (3, 14, 14, 24, 64, 3, 1),       # Cin not a multiple of the 16-wide channel chunk
    (2, 28, 28, 96, 128, 3, 1),
    (4, 7, 7, 832, 624, 1, 1),       # ragged last column tile
    (2, 13, 11, 48, 176, 3, 1),      # odd extents, rows past M in the last tile
    (1, 8, 8, 8, 200, 1, 1),         # one half-empty channel chunk
    (2, 10, 10, 24, 40, 3, 2),       # stride 2
    (2, 56, 56, 64, 192, 3, 1),
    (2, 14, 14, 512, 296, 1, 1),
    (1, 5, 5, 8, 8, 3, 1),           # fewer columns than one block
]


@pytest.mark.parametrize("nb", [1, 2])
@pytest.mark.parametrize("case", [(2, 8, 8, 16, 32), (2, 28, 28, 96, 128), (5, 7, 7, 160, 320), (2, 13, 11, 48, 176), (1, 9, 10, 16, 40),
                                  (1, 56, 56, 64, 192), (3, 14, 14, 32, 64), (9, 5, 6, 16, 48)])
def test_winograd_f4_on_bf16_matrix_cores_matches_bf16_rounding_oracle(case, nb, tuning_lib):
    """ds_conv_wino4_bf16x2 (the 16-bit configurations' 3x3 kernel: F(4x4, 3x3) of the bf16-ROUNDED operands, every
    Winograd-domain value as two bf16 pieces, three v_mfma_f32_32x32x16_bf16 per product) against the fp64 direct convolution of
    the bf16-rounded operands: forward with statistics about a pivot, dgrad plain and with the BatchNorm-sums epilogue from
    fp32 and from bf16 activation storage; both channel-block counts, border tiles, ragged groups.  Bound 5e-4 of max|ref|
    (measured 1e-4: two pieces carry 16 mantissa bits through transforms with constants up to 100), and 1e-2 against the
    exact convolution as for the other bf16 kernels."""
    ops = _ops()
    from tumblr_emotions_amd import _lib
    lib = _lib.load()
    N, H, W, Ci, Co = case
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.RandomState(11)
    x = rng.normal(size=(N, H, W, Ci))
    w = rng.normal(size=(3, 3, Ci, Co)) * 0.1
    ref = S.conv2d_same(_bf16_round(x), _bf16_round(w), 1)
    M = N * H * W
    xd, wd = dev(x), dev(w)
    P = lib.ds_conv_wino4_partials(N, H, W)
    assert lib.ds_debug_conv_wino4_set_nb(nb) == 0
    try:
        u2 = torch.empty(36 * Ci * Co, device="cuda")
        assert lib.ds_wino4_transform_weights_bf16x2(ops._p(wd), ops._p(u2), Ci, Co, 0, st) == 0
        z = torch.full((M, Co), float("nan"), device="cuda")
        stats = torch.zeros(2, Co, P, device="cuda")
        pivot = dev(rng.normal(size=Co) * 0.1)
        assert lib.ds_conv_wino4_bf16x2(ops._p(xd), ops._p(u2), ops._p(z), ops._p(stats), ops._p(pivot), None, ops.DS_DTYPE_F32,
                                        N, H, W, Ci, Ci, Co, Co, ops.DS_EPI_STATS, st) == 0
        torch.cuda.synchronize()
        zz = ref.reshape(M, Co)
        close(z, zz, 5e-4)
        close(z, S.conv2d_same(x, w, 1).reshape(M, Co), 1e-2)
        pv = pivot.cpu().numpy().astype(np.float64)
        close(stats[0].sum(1), (zz - pv).sum(0), 2e-3)
        close(stats[1].sum(1), ((zz - pv) ** 2).sum(0), 2e-3)
        if Co % 16 == 0:
            dy = rng.normal(size=ref.shape)
            dyd = dev(dy)
            ud = torch.empty(36 * Ci * Co, device="cuda")
            assert lib.ds_wino4_transform_weights_bf16x2(ops._p(wd), ops._p(ud), Ci, Co, 1, st) == 0
            dx = torch.zeros(M, Ci + 4, device="cuda")          # strided output rows
            assert lib.ds_conv_wino4_bf16x2(ops._p(dyd), ops._p(ud), ops._p(dx), None, None, None, ops.DS_DTYPE_F32,
                                            N, H, W, Co, Co, Ci, Ci + 4, 0, st) == 0
            torch.cuda.synchronize()
            dx_ref = S.conv2d_same_bwd_input(_bf16_round(dy), _bf16_round(w), (N, H, W, Ci), 1).reshape(-1, Ci)
            close(dx[:, :Ci], dx_ref, 5e-4)
            assert float(dx[:, Ci:].abs().max()) == 0.0
            yv = np.maximum(rng.normal(size=(M, Ci)), 0.0) * (rng.uniform(size=(M, Ci)) < 0.7)
            ypad = np.pad(yv, ((0, 0), (0, 4)), constant_values=5.0)
            for dt, yd in ((ops.DS_DTYPE_F32, dev(ypad)), (ops.DS_DTYPE_BF16, dev(ypad, torch.bfloat16))):
                y_seen = yd.float().cpu().numpy().astype(np.float64)[:, :Ci]
                sums = torch.full((2, Ci, P), float("nan"), device="cuda")
                dx2 = torch.zeros(M, Ci + 4, device="cuda")
                assert lib.ds_conv_wino4_bf16x2(ops._p(dyd), ops._p(ud), ops._p(dx2), ops._p(sums), None, ops._p(yd), dt,
                                                N, H, W, Co, Co, Ci, Ci + 4, ops.DS_EPI_BNSUMS, st) == 0
                torch.cuda.synchronize()
                assert torch.equal(dx2, dx)
                # the same launch reading dy from 16-bit storage (ds_conv_wino4_bf16x2_x16: the bf16 dz of ds_bn_bwd_apply_bf16):
                # the stored values are the ones the fp32 form rounds on load, so output and sums have the same bits
                dy16 = dyd.to(torch.bfloat16)
                sums16 = torch.full((2, Ci, P), float("nan"), device="cuda")
                dx3 = torch.zeros(M, Ci + 4, device="cuda")
                assert lib.ds_conv_wino4_bf16x2_x16(ops._p(dy16), ops._p(ud), ops._p(dx3), ops._p(sums16), None, ops._p(yd), dt,
                                                    N, H, W, Co, Co, Ci, Ci + 4, ops.DS_EPI_BNSUMS, st) == 0
                torch.cuda.synchronize()
                assert torch.equal(dx3, dx) and torch.equal(sums16, sums)
                gm = dx[:, :Ci].cpu().numpy().astype(np.float64) * (y_seen > 0)          # sums of what the kernel stored
                close(sums[0].sum(1), gm.sum(0), 2e-3)
                close(sums[1].sum(1), (gm * y_seen).sum(0), 2e-3)
    finally:
        lib.ds_debug_conv_wino4_set_nb(0)


@pytest.mark.parametrize("case", [(2, 7, 7, 160, 320, 4), (5, 7, 7, 320, 160, 3), (2, 14, 14, 96, 208, 2), (1, 14, 14, 208, 96, 13), (3, 5, 6, 48, 44, 2),
                                  (2, 8, 8, 32, 32, 2)])
def test_winograd_f4_with_the_reduction_split_over_workgroups(case):
    """ds_conv_wino4_splitk (round 6; small per-GPU batches, image_model/inception_v1.py:122-247): the reduction channels of an
    F(4x4, 3x3) launch in `splits` slices, one workgroup each, their partial outputs added by a second launch that also runs the
    DS_EPI_STATS / DS_EPI_BNSUMS epilogue.  Against the fp64 convolution at the unsplit kernel's bound (2e-4 of max|ref|) and
    against ds_conv_wino4 itself (fp32 summation order: 2e-5); uneven slices, slice counts up to Cin / 16, border tiles,
    strided output rows, y in fp32 and bf16 storage."""
    ops = _ops()
    from tumblr_emotions_amd import _lib
    lib = _lib.load()
    N, H, W, Ci, Co, S_ = case
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.RandomState(21)
    x = rng.normal(size=(N, H, W, Ci))
    w = rng.normal(size=(3, 3, Ci, Co)) * 0.1
    ref = S.conv2d_same(x, w, 1)
    M = N * H * W
    xd, wd = dev(x), dev(w)
    P0, P = lib.ds_conv_wino4_partials(N, H, W), lib.ds_conv_wino4_splitk_partials(N, H, W)
    wsb = lib.ds_conv_wino4_splitk_workspace(N, H, W, max(Ci, Co), S_)
    assert wsb == S_ * M * max(Ci, Co) * 4 and P >= 1
    ws = torch.full((wsb // 4,), float("nan"), device="cuda")
    u = torch.empty(36 * Ci * Co, device="cuda")
    assert lib.ds_wino4_transform_weights(ops._p(wd), ops._p(u), Ci, Co, 0, st) == 0
    pivot = dev(rng.normal(size=Co) * 0.1)
    z0, z = torch.zeros(M, Co + 4, device="cuda"), torch.zeros(M, Co + 4, device="cuda")
    st0, st1 = torch.zeros(2, Co, P0, device="cuda"), torch.full((2, Co, P), float("nan"), device="cuda")
    assert lib.ds_conv_wino4(ops._p(xd), ops._p(u), ops._p(z0), ops._p(st0), ops._p(pivot), None, N, H, W, Ci, Ci, Co, Co + 4,
                             ops.DS_EPI_STATS, st) == 0
    assert lib.ds_conv_wino4_splitk(ops._p(xd), ops._p(u), ops._p(z), ops._p(st1), ops._p(pivot), None, ops.DS_DTYPE_F32, N, H, W, Ci, Ci,
                                    Co, Co + 4, ops.DS_EPI_STATS, S_, ops._p(ws), wsb, st) == 0
    torch.cuda.synchronize()
    zz = ref.reshape(M, Co)
    close(z[:, :Co], zz, 2e-4)
    close(z[:, :Co], z0[:, :Co].cpu().numpy().astype(np.float64), 2e-5)
    assert float(z[:, Co:].abs().max()) == 0.0
    pv = pivot.cpu().numpy().astype(np.float64)
    close(st1[0].sum(1), (zz - pv).sum(0), 1e-3)
    close(st1[1].sum(1), ((zz - pv) ** 2).sum(0), 1e-3)
    close(st1[1].sum(1), st0[1].sum(1).cpu().numpy().astype(np.float64), 1e-4)
    # plain launch (no epilogue) and the input gradient with the BatchNorm-sums epilogue
    z2 = torch.zeros(M, Co, device="cuda")
    assert lib.ds_conv_wino4_splitk(ops._p(xd), ops._p(u), ops._p(z2), None, None, None, ops.DS_DTYPE_F32, N, H, W, Ci, Ci, Co, Co, 0, S_,
                                    ops._p(ws), wsb, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(z2, z[:, :Co])
    if Co % 16 == 0 and S_ <= Co // 16:
        dy = rng.normal(size=ref.shape)
        dyd = dev(dy)
        ud = torch.empty(36 * Ci * Co, device="cuda")
        assert lib.ds_wino4_transform_weights(ops._p(wd), ops._p(ud), Ci, Co, 1, st) == 0
        dx_ref = S.conv2d_same_bwd_input(dy, w, (N, H, W, Ci), 1).reshape(-1, Ci)
        yv = np.maximum(rng.normal(size=(M, Ci)), 0.0) * (rng.uniform(size=(M, Ci)) < 0.7)
        for dt, yd in ((ops.DS_DTYPE_F32, dev(yv)), (ops.DS_DTYPE_BF16, dev(yv, torch.bfloat16))):
            y_seen = yd.float().cpu().numpy().astype(np.float64)
            sums = torch.full((2, Ci, P), float("nan"), device="cuda")
            dx = torch.zeros(M, Ci, device="cuda")
            assert lib.ds_conv_wino4_splitk(ops._p(dyd), ops._p(ud), ops._p(dx), ops._p(sums), None, ops._p(yd), dt, N, H, W, Co, Co, Ci, Ci,
                                            ops.DS_EPI_BNSUMS, S_, ops._p(ws), wsb, st) == 0
            torch.cuda.synchronize()
            close(dx, dx_ref, 2e-4)
            gm = dx.cpu().numpy().astype(np.float64) * (y_seen > 0)
            close(sums[0].sum(1), gm.sum(0), 1e-4)
            close(sums[1].sum(1), (gm * y_seen).sum(0), 1e-4)
    # the launch-time model: never at the headline batch's big maps, yes for the 7x7 input gradient of a small batch
    assert lib.ds_conv_wino4_splitk_choose(256, 28, 28, 128, 192) == 1
    assert lib.ds_conv_wino4_splitk_choose(32, 7, 7, 320, 160) >= 2


@pytest.mark.parametrize("case", BF16D_CASES)
def test_conv_bf16_register_direct_forward_dgrad_match_oracle(case):
    """ds_conv_bf16 (register-direct A, pre-converted weights): forward with BatchNorm statistics about a pivot and
    the dgrad of the same geometry; same two gates as the LDS-staged bf16 kernel.  The converted weight tensor
    itself is checked element for element against numpy's rounding."""
    ops = _ops()
    N, H, W, Ci, Co, k, s = case
    rng = np.random.RandomState(5)
    x = rng.normal(size=(N, H, W, Ci))
    w = rng.normal(size=(k, k, Ci, Co)) * 0.1
    ref_exact = S.conv2d_same(x, w, s)
    ref_round = S.conv2d_same(_bf16_round(x), _bf16_round(w), s)
    xd, wd = dev(x), dev(w)
    wb = torch.empty(ops.weights_bf16_bytes(Ci, Co, k * k, 0), dtype=torch.uint8, device="cuda")
    ops.weights_to_bf16(ops._p(wd), wb, Ci, Co, k * k, 0)
    # layout check: wb[it][col][16], it = chunk * taps + tap
    chunks, ncols = (Ci + 15) // 16, (Co + 31) // 32 * 32
    got = wb.view(torch.bfloat16).float().cpu().numpy().reshape(chunks, k * k, ncols, 16)
    want = np.zeros((chunks * 16, k * k, ncols))
    want[:Ci, :, :Co] = _bf16_round(w).reshape(k * k, Ci, Co).transpose(1, 0, 2)
    want = want.reshape(chunks, 16, k * k, ncols).transpose(0, 2, 3, 1)
    assert np.array_equal(got, want.astype(np.float32))
    plan = ops.Bf16Plan(N, H, W, Ci, Ci, k, s, Co, Co, flags=ops.DS_EPI_STATS)
    M = plan.M
    z = torch.empty(M, Co, device="cuda")
    stats = torch.zeros(2, Co, plan.partials, device="cuda")
    pivot = dev(rng.normal(size=Co))
    plan.run(ops._p(xd), ops._p(wb), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot))
    torch.cuda.synchronize()
    close(z, ref_round.reshape(M, Co), 2e-4)
    close(z, ref_exact.reshape(M, Co), 1e-2)
    u = ref_round.reshape(M, Co) - pivot.double().cpu().numpy()
    close(stats[0].sum(1), u.sum(0), 2e-3)
    close(stats[1].sum(1), (u ** 2).sum(0), 2e-3)
    if s == 1 and Co % 8 == 0:
        dy = rng.normal(size=ref_exact.shape)
        wbt = torch.empty(ops.weights_bf16_bytes(Ci, Co, k * k, 1), dtype=torch.uint8, device="cuda")
        ops.weights_to_bf16(ops._p(wd), wbt, Ci, Co, k * k, 1)
        g = ops.Bf16Plan(N, H, W, Co, Co, k, 1, Ci, Ci)
        dx = torch.empty(g.M, Ci, device="cuda")
        dyd = dev(dy)
        g.run(ops._p(dyd), ops._p(wbt), ops._p(dx))
        torch.cuda.synchronize()
        dref = S.conv2d_same_bwd_input(_bf16_round(dy), _bf16_round(w), (N, H, W, Ci), 1).reshape(-1, Ci)
        close(dx, dref, 2e-4)
        close(dx, S.conv2d_same_bwd_input(dy, w, (N, H, W, Ci), 1).reshape(-1, Ci), 1e-2)

        def run(flags, ldmask, mask_dtype, out, mask):
            g2 = ops.Bf16Plan(N, H, W, Co, Co, k, 1, Ci, Ci, flags=flags)
            g2.d.ldmask, g2.d.mask_dtype = ldmask, mask_dtype
            P = (g2.M + 127) // 128
            sums = torch.full((2, Ci, P), float("nan"), device="cuda")
            g2.run(ops._p(dyd), ops._p(wbt), ops._p(out), stats=ops._p(sums), mask=ops._p(mask))
            return sums, P
        _check_accum_bnsums_epilogue(run, dref, g.M, Ci, rng, 2e-4)


def test_conv_bf16_folded_stem_and_epilogues():
    """The 7x7/2 stem through its folded 4-channel input, and the bias / relu / accumulate / mask epilogues, on
    the bf16 matrix pipe."""
    ops = _ops()
    rng = np.random.RandomState(4)
    N, H, Co = 2, 64, 64
    x = rng.uniform(-1, 1, size=(N, H, H, 3))
    w = rng.normal(size=(7, 7, 3, Co)) * 0.1
    ref = S.conv2d_same(_bf16_round(x), _bf16_round(w), 2)
    x4 = np.zeros((N, H, H, 4)); x4[..., :3] = x
    w4 = np.zeros((7, 7, 4, Co)); w4[:, :, :3] = w
    plan = ops.ConvPlan(N, H, H, 7 * 4, 4, 7, 1, 2, Co, Co, 28 * Co, 1, Co, fold_cin=4, dtype=ops.DS_DTYPE_BF16)
    z = torch.empty(plan.M, Co, device="cuda")
    xd, wd = dev(x4), dev(w4)
    plan.run(ops._p(xd), ops._p(wd), ops._p(z))
    torch.cuda.synchronize()
    close(z, ref.reshape(plan.M, Co), 2e-4)
    # GEMM with bias + relu, then accumulate + mask
    M, K, Nn = 300, 72, 40
    a, b, bias = rng.normal(size=(M, K)), rng.normal(size=(K, Nn)) * 0.2, rng.normal(size=Nn)
    prev, mask = rng.normal(size=(M, Nn)), (rng.uniform(size=(M, Nn)) < 0.5).astype(np.float64)
    ad, bd, biasd, maskd = dev(a), dev(b), dev(bias), dev(mask)
    core = _bf16_round(a) @ _bf16_round(b)
    out = torch.empty(M, Nn, device="cuda")
    ops.gemm_plan(M, K, Nn, K, Nn, Nn, flags=ops.DS_EPI_BIAS | ops.DS_EPI_RELU, dtype=ops.DS_DTYPE_BF16).run(
        ops._p(ad), ops._p(bd), ops._p(out), bias=ops._p(biasd))
    close(out, np.maximum(core + bias, 0), 2e-4)
    acc = dev(prev)
    ops.gemm_plan(M, K, Nn, K, Nn, Nn, flags=ops.DS_EPI_ACCUM | ops.DS_EPI_MASK, ldmask=Nn, dtype=ops.DS_DTYPE_BF16).run(
        ops._p(ad), ops._p(bd), ops._p(acc), mask=ops._p(maskd))
    close(acc, (core + prev) * mask, 2e-4)


@pytest.mark.parametrize("case", [(8, 6, 32), (37, 9, 64), (64, 12, 128), (256, 32, 512), (33, 5, 1024), (5, 50, 256)])
def test_lstm_sequence_kernels_match_oracle(case):
    """ds_lstm_seq_fwd / ds_lstm_seq_bwd (whole recurrence in one persistent launch per direction) against the
    NumPy BasicLSTMCell + dynamic_rnn restatement: every h_t, c_t, the activations, the last valid output, and
    every per-step gate gradient; ragged batch (B not a multiple of the 32-row groups), sequence lengths 1 and
    T, every supported hidden size incl. the 8-wave H = 1024 variant and BASELINE's (256, 32, 512)."""
    ops = _ops()
    B, T, H = case
    rng = np.random.RandomState(B + T + H)
    pre = rng.normal(size=(T, B, 4 * H)) * 0.7                     # x_t Wx + b, already hoisted
    wh = rng.normal(size=(H, 4 * H)) * (0.5 / np.sqrt(H))
    seq = rng.randint(1, T + 1, size=B).astype(np.int64)
    seq[0], seq[-1] = 1, T
    dh_last = rng.normal(size=(B, H))
    # oracle: feed the pre-activations through an identity input block of the TF kernel
    kernel = np.concatenate([np.eye(4 * H), wh], axis=0)
    outs, h_last, cache = S.lstm_forward(pre.transpose(1, 0, 2), seq, kernel, np.zeros(4 * H), keep_cache=True)
    _, _, dzs = S.lstm_backward(dh_last, seq, kernel, cache, return_dz=True)

    gates, whd = dev(pre), dev(wh)
    h = torch.zeros(T + 1, B, H, device="cuda")
    c = torch.zeros(T + 1, B, H, device="cuda")
    seqd = torch.from_numpy(seq).cuda()
    ws = torch.zeros(max(ops.lstm_seq_workspace(B, H) // 4, 4), dtype=torch.int32, device="cuda")
    assert ops.lstm_seq_supported(B, H)
    ops.lstm_seq_fwd(gates, ops._p(whd), 4 * H, h, c, seqd, T, B, H, S.FORGET_BIAS, ws)
    torch.cuda.synchronize()
    ops.lstm_seq_status(ws, B)
    close(h[T], h_last)
    hn, cn, an = h.cpu().numpy(), c.cpu().numpy(), gates.cpu().numpy()
    for t in range(T):
        live = t < seq
        q = cache[t]
        assert np.abs(hn[t + 1][live] - outs[live, t]).max() <= 2e-4, t
        c_ref = q["c_prev"] * q["sf"] + q["si"] * q["tj"]
        assert np.abs(cn[t + 1][live] - c_ref[live]).max() <= 2e-4, t
        acts = np.concatenate([q["si"], q["tj"], q["sf"], q["so"]], axis=1)
        assert np.abs(an[t] - acts).max() <= 2e-4, t
        if (~live).any():            # copy-through past seq_len
            np.testing.assert_array_equal(hn[t + 1][~live], hn[t][~live])
            np.testing.assert_array_equal(cn[t + 1][~live], cn[t][~live])
    dg = torch.full((T, B, 4 * H), float("nan"), device="cuda")
    dhd = dev(np.pad(dh_last, ((0, 0), (0, 8))))[:, :H]            # a strided view: ld_dh != H
    ops.lstm_seq_bwd(gates, ops._p(whd), 4 * H, c, dhd, dhd.stride(0), seqd, T, B, H, dg, ws)
    torch.cuda.synchronize()
    ops.lstm_seq_status(ws, B)
    dgn = dg.cpu().numpy()
    scale = max(np.abs(d).max() for d in dzs)
    for t in range(T):
        assert np.abs(dgn[t] - dzs[t]).max() <= 3e-4 * scale, t
    # two runs are bit-identical (fixed reduction order, no atomics on data)
    dg2 = torch.empty_like(dg)
    ops.lstm_seq_bwd(gates, ops._p(whd), 4 * H, c, dhd, dhd.stride(0), seqd, T, B, H, dg2, ws)
    torch.cuda.synchronize()
    assert torch.equal(dg, dg2)
    # `rows`: R row groups per workgroup (fewer, longer workgroups) changes scheduling only: among the R > 1 launches
    # the forward pass gives the same bits, the backward pass the same values to the last bit or two (hipcc contracts
    # the gate derivatives' multiply-adds differently in the R > 1 instantiations of the small hidden sizes); a last
    # workgroup with fewer than R row groups is included.  Against the rows = 1 launch above the agreement is to
    # rounding only where that one ran 16-row groups (small batches, H = 64 ... 512: the 16x16x4 MFMA sums the
    # reduction in another order than the 32x32x2 one).
    ref = None
    for rows in (2, 4, 8):
        g2, h2, c2 = dev(pre), torch.zeros_like(h), torch.zeros_like(c)
        ops.lstm_seq_fwd(g2, ops._p(whd), 4 * H, h2, c2, seqd, T, B, H, S.FORGET_BIAS, ws, rows=rows)
        dg3 = torch.empty_like(dg)
        ops.lstm_seq_bwd(g2, ops._p(whd), 4 * H, c2, dhd, dhd.stride(0), seqd, T, B, H, dg3, ws, rows=rows)
        torch.cuda.synchronize()
        ops.lstm_seq_status(ws, B)
        if ref is None:
            ref = (h2, c2, g2, dg3)
        assert torch.equal(h2, ref[0]) and torch.equal(c2, ref[1]) and torch.equal(g2, ref[2]), rows
        assert float((dg3 - ref[3]).abs().max()) <= 4e-7 * float(dg.abs().max()), rows
        for got, want in ((h2, h), (c2, c), (g2, gates)):
            assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())), rows
        assert float((dg3 - dg).abs().max()) <= 2e-6 * float(dg.abs().max()), rows


@pytest.mark.parametrize("case", [(256, 32, 512, 1), (256, 32, 512, 4), (256, 32, 512, 8), (200, 32, 512, 8), (160, 32, 512, 4), (64, 12, 128, 1), (37, 9, 64, 2), (100, 20, 256, 1)])
def test_lstm_sequence_skips_masked_steps_to_the_bit_on_length_sorted_batches(case):
    """ds_seq_sort_desc + ds_permute_rows + DS_LSTM_SKIP_MASKED (round 6): the batch in descending order of length, every 32- /
    16-row group stopping after its longest row -- h at every step, c at every step, the last valid output and every per-step
    gate gradient have the bits of the plain launch on the same (sorted) batch; un-sorting the last output reproduces the
    plain launch on the ORIGINAL batch sample by sample (rows are independent).  Lengths include 1, T, a whole group of equal
    lengths and groups that end long before T."""
    ops = _ops()
    from tumblr_emotions_amd import _lib
    B, T, H, rows = case
    rng = np.random.RandomState(B + T + H)
    seq = rng.randint(1, T + 1, size=B).astype(np.int64)
    seq[0], seq[-1] = 1, T
    seq[B // 2:B // 2 + min(40, B // 4)] = max(1, T // 3)
    seqd = torch.from_numpy(seq).cuda()
    perm = torch.empty(B, dtype=torch.int32, device="cuda")
    lens = torch.empty(B, dtype=torch.int64, device="cuda")
    ops.seq_sort_desc(seqd, B, T, perm, lens)
    torch.cuda.synchronize()
    order = np.array(sorted(range(B), key=lambda i: (-seq[i], i)))
    assert np.array_equal(perm.cpu().numpy(), order) and np.array_equal(lens.cpu().numpy(), seq[order])
    pre = torch.from_numpy((rng.normal(size=(T, B, 4 * H)) * 0.7).astype(np.float32)).cuda()      # original order
    pre_s = pre[:, perm.long()].contiguous()
    ids = torch.from_numpy(rng.randint(0, 99, size=(B, T)).astype(np.int64)).cuda()
    ids_s = torch.empty_like(ids)
    ops.permute_rows(ids, ids_s, perm, B, T, gather=True)
    assert torch.equal(ids_s, ids[perm.long()])
    wh = dev(rng.normal(size=(H, 4 * H)) * (0.5 / np.sqrt(H)))
    dh = torch.from_numpy(rng.normal(size=(B, H)).astype(np.float32)).cuda()
    dh_s = torch.empty_like(dh)
    ops.permute_rows(dh, dh_s, perm, B, H, gather=True)
    ws = torch.zeros(max(ops.lstm_seq_workspace(B, H) // 4, 4), dtype=torch.int32, device="cuda")

    def run(gates, sl, dhl, flag):
        g = gates.clone()
        h, c = torch.zeros(T + 1, B, H, device="cuda"), torch.zeros(T + 1, B, H, device="cuda")
        ops.lstm_seq_fwd(g, ops._p(wh), 4 * H, h, c, sl, T, B, H, S.FORGET_BIAS, ws, rows=rows | flag)
        dg = torch.full((T, B, 4 * H), float("nan"), device="cuda")
        ops.lstm_seq_bwd(g, ops._p(wh), 4 * H, c, dhl, dhl.stride(0), sl, T, B, H, dg, ws, rows=rows | flag)
        torch.cuda.synchronize()
        ops.lstm_seq_status(ws, B)
        return h, c, dg

    h0, c0, dg0 = run(pre_s, lens, dh_s, 0)
    h1, c1, dg1 = run(pre_s, lens, dh_s, _lib.DS_LSTM_SKIP_MASKED)
    assert torch.equal(h0, h1) and torch.equal(c0, c1) and torch.equal(dg0, dg1)
    hu, _, dgu = run(pre, seqd, dh, 0)                              # the plain launch on the ORIGINAL order
    back = torch.empty(B, H, device="cuda")
    ops.permute_rows(h1[T], back, perm, B, H, gather=False)
    torch.cuda.synchronize()
    if B % 32 == 0 or rows == 1:          # (same MFMA form for every row: a ragged last group may change between 16- and 32-row groups)
        assert torch.equal(back, hu[T])
        assert torch.equal(dg1, dgu[:, perm.long()])
    else:
        assert float((back - hu[T]).abs().max()) <= 2e-6


def test_lstm_sequence_is_reentrant_across_streams_and_row_settings():
    """no global mutable state.  Two text towers with DIFFERENT `rows` settings run at the same time on
    two streams (each with its own workspace) and give the bits each gives alone; the forward and the backward
    launch keep separate, sticky error words (a backward launch no longer wipes the forward's)."""
    ops = _ops()
    from tumblr_emotions_amd import _lib
    lib = _lib.load()
    B, T, H = 96, 12, 128
    rng = np.random.RandomState(11)
    wh = dev(rng.normal(size=(H, 4 * H)) * (0.5 / np.sqrt(H)))
    seqd = torch.from_numpy(rng.randint(1, T + 1, size=B).astype(np.int64)).cuda()
    dh_last = dev(rng.normal(size=(B, H)))

    def run(pre, rows, ws):
        g, h, c = pre.clone(), torch.zeros(T + 1, B, H, device="cuda"), torch.zeros(T + 1, B, H, device="cuda")
        dg = torch.empty(T, B, 4 * H, device="cuda")
        ops.lstm_seq_fwd(g, ops._p(wh), 4 * H, h, c, seqd, T, B, H, S.FORGET_BIAS, ws, rows=rows)
        ops.lstm_seq_bwd(g, ops._p(wh), 4 * H, c, dh_last, H, seqd, T, B, H, dg, ws, rows=rows)
        return h, dg

    pres = [dev(rng.normal(size=(T, B, 4 * H)) * 0.7) for _ in range(2)]
    wss = [torch.zeros(max(ops.lstm_seq_workspace(B, H) // 4, 4), dtype=torch.int32, device="cuda") for _ in range(2)]
    alone = [run(pres[i], (1, 2)[i], wss[i]) for i in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        both = []
        for i in range(2):
            streams[i].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[i]):
                both.append(run(pres[i], (1, 2)[i], wss[i]))
        torch.cuda.synchronize()
        for i in range(2):
            ops.lstm_seq_status(wss[i], B)
            assert torch.equal(both[i][0], alone[i][0]) and torch.equal(both[i][1], alone[i][1]), (rep, i)
    # sticky, separate error words: poke the forward word, run a backward launch, the status still reports bit 0
    nrg = 65 * ((B + 15) // 16)      # the two error words sit behind the two directions' blocks of 65 * ceil(B / 16) words each
    wss[0][2 * nrg] = 1
    run(pres[0], 1, wss[0])
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="forward"):
        ops.lstm_seq_status(wss[0], B)
    # ... and a reported failure is cleared by the read: the next, healthy step is not blamed for it
    assert lib.ds_lstm_seq_status(ops._p(wss[0]), B) == 0
    wss[0][2 * nrg] = 1
    wss[0][2 * nrg + 1] = 1
    assert lib.ds_lstm_seq_status(ops._p(wss[0]), B) == 3
    assert lib.ds_lstm_seq_status(ops._p(wss[0]), B) == 0
    # rows outside {1, 2, 4, 8} is an argument error, not a silent default
    with pytest.raises(RuntimeError, match="rows"):
        ops.lstm_seq_fwd(pres[0].clone(), ops._p(wh), 4 * H, torch.zeros(T + 1, B, H, device="cuda"),
                         torch.zeros(T + 1, B, H, device="cuda"), seqd, T, B, H, S.FORGET_BIAS, wss[1], rows=3)


def test_text_tower_persistent_and_stepwise_paths_agree():
    """TextTowerEngine with the persistent recurrence against the same engine forced onto the step-wise pair
    (one GEMM + one cell launch per step): same h_last, same gradients to rounding."""
    from tumblr_emotions_amd.net import SentimentNet
    from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
    res = []
    for persistent in (True, False):
        net = SentimentNet(mode="text", nb_emotions=15, rnn_size=64, vocab_size=80, embedding_dim=24, post_size=14)
        net.text.persistent = persistent
        net.initialize(seed=5)
        batch = to_device(synthetic_batch_numpy(40, 14, 80, seed=2, with_images=False))
        net.train_step(batch, 1e-3)
        torch.cuda.synchronize()
        assert net.text.use_seq == persistent
        if persistent:
            ops = _ops()
            ops.lstm_seq_status(net.text.seq_ws, 40)
        res.append((net.logits.clone(), net.store.grad.clone()))
    assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-5
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-5 * max(1.0, float(res[1][1].abs().max()))


def test_batch_norm_forward_backward_with_segments():
    ops = _ops()
    rng = np.random.RandomState(7)
    M, Cc = 1000, 176                                       # fused 1x1 of Mixed_3b: 64 | 96 | 16
    z = rng.normal(0.5, 2.0, size=(M, Cc))
    beta = rng.normal(size=Cc) * 0.3
    y_ref, mean, var, xhat, rstd = S.batch_norm_train(z, beta)
    y_ref = S.relu(y_ref)
    zd, bd = dev(z), dev(beta)
    # forward statistics through the conv epilogue path is covered elsewhere; here feed exact partials
    stats = torch.stack([zd.sum(0), (zd * zd).sum(0)]).reshape(2, Cc, 1).contiguous()
    mean_d, rstd_d, shift_d = (torch.empty(Cc, device="cuda") for _ in range(3))
    mm, mv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    ops.bn_finalize(stats, 1, M, Cc, bd, S.BN_EPS, S.BN_DECAY, mean_d, rstd_d, shift_d, mm, mv)
    close(mean_d, mean, 1e-4)
    close(rstd_d, rstd, 1e-4)
    close(mm, 0.0003 * mean, 1e-3)
    close(mv, 0.9997 + 0.0003 * var, 1e-5)
    concat = torch.zeros(M, 256, device="cuda")
    r1 = torch.zeros(M, 96, device="cuda")
    r2 = torch.zeros(M, 16, device="cuda")
    segs = ops.make_segments([(0, 64, concat.data_ptr(), 256), (64, 160, r1.data_ptr(), 96),
                              (160, 176, r2.data_ptr(), 16)])
    ops.bn_apply_relu(zd, M, Cc, rstd_d, shift_d, segs)
    close(concat[:, :64], y_ref[:, :64])
    close(r1, y_ref[:, 64:160])
    close(r2, y_ref[:, 160:])
    assert float(concat[:, 64:].abs().max()) == 0.0
    # backward
    dy = rng.normal(size=(M, Cc))
    g = dy * (y_ref > 0)
    dz_ref, dbeta_ref = S.batch_norm_train_bwd(g, xhat, rstd)
    dyc, dr1, dr2 = dev(np.pad(dy[:, :64], ((0, 0), (0, 192)))), dev(dy[:, 64:160]), dev(dy[:, 160:])
    dsegs = ops.make_segments([(0, 64, dyc.data_ptr(), 256), (64, 160, dr1.data_ptr(), 96),
                               (160, 176, dr2.data_ptr(), 16)])
    P = ops.bn_bwd_partials(M, Cc)
    part = torch.empty(2, Cc, P, device="cuda")
    dbeta, coef = torch.empty(Cc, device="cuda"), torch.empty(2, Cc, device="cuda")
    ops.bn_bwd_reduce(zd, dsegs, M, Cc, mean_d, rstd_d, shift_d, part)
    ops.bn_bwd_finalize(part, P, M, Cc, dbeta, coef)
    ops.bn_bwd_apply(zd, dsegs, M, Cc, mean_d, rstd_d, shift_d, coef, zd)        # in place over z
    torch.cuda.synchronize()
    close(dbeta, dbeta_ref, 1e-4)
    close(zd, dz_ref, 3e-4)


@pytest.mark.parametrize("geometry", [(8, 28, 28, 192, 96, 1), (2, 56, 56, 64, 192, 3), (2, 370, 370, 16, 64, 3)])
def test_batch_norm_statistics_of_channels_with_mean_50_sigma(geometry):
    """tf.nn.moments (slim.batch_norm, slim/nets/inception_utils.py:48-70) averages squared differences, so it
    is exact for |mean| >> std; a one-pass E[z^2] - E[z]^2 over fp32 partial sums is not.  Conv outputs with
    |mean| = 50 std per channel (a constant input plane behind large weights, as a warm-started checkpoint can
    produce): mean / variance / rstd from the conv epilogue + ds_bn_finalize against the fp64 oracle, with the
    pivot the engine would pass (the previous step's batch mean: here the true mean off by 0.5 std), and the
    normalised activations and BN backward on top of those statistics.  The last geometry runs the persistent
    launch path (one workgroup sums several row tiles)."""
    ops = _ops()
    N, H, W, Ci, Co, k = geometry
    rng = np.random.RandomState(12)
    x = rng.normal(size=(N, H, W, Ci))
    x[..., 0] = 1.0                                           # constant plane: carries the channel offsets
    w = rng.normal(size=(k, k, Ci, Co)) * 0.1
    w[:, :, 0, :] = 0.0
    z0 = S.conv2d_same(x, w, 1)
    sig = z0.reshape(-1, Co).std(axis=0)
    w[k // 2, k // 2, 0, :] = 50.0 * sig * rng.choice([-1.0, 1.0], size=Co)       # |mean| = 50 sigma
    x32, w32 = x.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64)
    z = S.conv2d_same(x32, w32, 1)
    M = N * H * W
    beta = rng.normal(size=Co) * 0.3
    y_ref, mean, var, xhat, rstd = S.batch_norm_train(z.reshape(M, Co), beta)
    assert np.abs(mean / np.sqrt(var)).min() > 30
    xd, wd, bd = dev(x), dev(w), dev(beta)
    plan = ops.ConvPlan(N, H, W, Ci, Ci, k, k, 1, Co, Co, Ci * Co, 1, Co, flags=ops.DS_EPI_STATS)
    zd = torch.empty(M, Co, device="cuda")
    stats = torch.zeros(2, Co, plan.partials, device="cuda")
    res = {}
    for label, pivot in (("pivot", dev(mean + 0.5 * np.sqrt(var))), ("no pivot", None)):
        mean_d, rstd_d, shift_d = (torch.empty(Co, device="cuda") for _ in range(3))
        mv = torch.zeros(Co, device="cuda")
        plan.run(ops._p(xd), ops._p(wd), ops._p(zd), stats=ops._p(stats), pivot=ops._p(pivot))
        ops.bn_finalize(stats, plan.partials, M, Co, bd, S.BN_EPS, 0.0, mean_d, rstd_d, shift_d, None, mv, pivot=pivot)
        torch.cuda.synchronize()
        res[label] = (np.abs(rstd_d.cpu().numpy() / rstd - 1).max(), np.abs(mv.cpu().numpy() / var - 1).max())
        if label == "pivot":
            keep = (mean_d.clone(), rstd_d.clone(), shift_d.clone())
            close(mean_d, mean, 1e-6)
            assert res[label][0] <= 2e-4 and res[label][1] <= 2e-4, res
    print("rstd / variance relative error, |mean| = 50 sigma:", res)
    mean_d, rstd_d, shift_d = keep
    out = torch.empty(M, Co, device="cuda")
    ops.bn_apply_relu(zd, M, Co, rstd_d, shift_d, ops.make_segments([(0, Co, out.data_ptr(), Co)]))
    close(out, S.relu(y_ref), 2e-4)
    dy = rng.normal(size=(M, Co))
    dz_ref, dbeta_ref = S.batch_norm_train_bwd(dy * (y_ref > 0), xhat, rstd)
    dyd = dev(dy)
    dsegs = ops.make_segments([(0, Co, dyd.data_ptr(), Co)])
    P = ops.bn_bwd_partials(M, Co)
    part = torch.empty(2, Co, P, device="cuda")
    dbeta, coef = torch.empty(Co, device="cuda"), torch.empty(2, Co, device="cuda")
    ops.bn_bwd_reduce(zd, dsegs, M, Co, mean_d, rstd_d, shift_d, part)
    ops.bn_bwd_finalize(part, P, M, Co, dbeta, coef)
    ops.bn_bwd_apply(zd, dsegs, M, Co, mean_d, rstd_d, shift_d, coef, zd)
    torch.cuda.synchronize()
    # a pre-activation within fp32 rounding of 0 may take the other ReLU branch: compare where |y| is resolved
    resolved = np.abs(y_ref) > 1e-3
    err = np.abs(zd.cpu().numpy() - dz_ref) * resolved
    assert err.max() <= 5e-4 * np.abs(dz_ref).max(), err.max()


@pytest.mark.parametrize("case", [(3, 2, 9, "SAME"), (3, 1, 7, "SAME"), (2, 2, 14, "VALID"), (3, 2, 112, "SAME")])
def test_max_pool_forward_backward(case):
    ops = _ops()
    k, s, H, mode = case
    rng = np.random.RandomState(8)
    N, Cc = 2, 24
    x = rng.normal(size=(N, H, H, Cc))
    ref = S.max_pool(x, k, s, mode)
    OH = ref.shape[1]
    xd = dev(x)
    y = torch.empty(N, OH, OH, Cc, device="cuda")
    am = torch.empty(N, OH, OH, Cc, dtype=torch.uint8, device="cuda")
    ops.maxpool_fwd(xd, y, am, N, H, H, Cc, k, s, mode)
    close(y, ref, 1e-6)
    dy = rng.normal(size=ref.shape)
    dx_ref = S.max_pool_bwd(x, dy, k, s, mode)
    base = rng.normal(size=x.shape)
    dx = dev(base)
    ops.maxpool_bwd(dev(dy), am, dx, True, N, H, H, Cc, k, s, mode)
    torch.cuda.synchronize()
    close(dx, base + dx_ref, 1e-6)


@pytest.mark.parametrize("case", [(2, 17, 24, 2), (3, 12, 64, 2), (2, 9, 20, 1)])
def test_max_pool_with_deferred_batch_norm_relu(case):
    """ds_maxpool_bn_relu_fwd: relu(rstd*maxpool(z)+shift) == maxpool(relu(rstd*z+shift)) (rstd > 0), and the
    gradient routed through its arg-max, masked by the ReLU of the pre-pool activation, equals the gradient of
    the unfused pair BatchNorm-ReLU -> MaxPool."""
    ops = _ops()
    N, H, Cc, s = case
    rng = np.random.RandomState(16)
    z = rng.normal(size=(N, H, H, Cc))
    rstd = rng.uniform(0.5, 2.0, size=Cc)
    shift = rng.normal(size=Cc) * 0.5
    y_full = np.maximum(z * rstd + shift, 0)
    ref = S.max_pool(y_full, 3, s, "SAME")
    OH = ref.shape[1]
    zd = dev(z)
    y = torch.empty(N, OH, OH, Cc, device="cuda")
    am = torch.empty(N, OH, OH, Cc, dtype=torch.uint8, device="cuda")
    ops.maxpool_bn_relu_fwd(zd, dev(rstd), dev(shift), y, am, N, H, H, Cc, 3, s)
    torch.cuda.synchronize()
    close(y, ref)
    dy = rng.normal(size=ref.shape)
    dx = torch.zeros(N, H, H, Cc, device="cuda")
    ops.maxpool_bwd(dev(dy), am, dx, False, N, H, H, Cc, 3, s, "SAME")
    torch.cuda.synchronize()
    got = dx.cpu().numpy() * (y_full > 0)               # what BatchNorm backward keeps of it
    want = S.max_pool_bwd(y_full, dy, 3, s, "SAME") * (y_full > 0)
    assert np.abs(got - want).max() <= 1e-6


@pytest.mark.parametrize("case", [(24, 28, 28, 192, 32, False, True), (24, 28, 28, 256, 64, True, True), (96, 14, 14, 480, 64, True, True),
                                  (96, 14, 14, 528, 128, True, True), (384, 7, 7, 832, 128, False, True), (3, 9, 11, 48, 96, True, False),
                                  (2, 5, 32, 64, 40, False, False), (1, 1, 1, 32, 32, True, False)])
def test_branch3_max_pool_formed_on_load_equals_pool_then_conv(case):
    """ds_conv_desc.pool_argmax (an Inception block's Branch_3, image_model/inception_v1.py:94-95: max_pool2d 3x3 / 1 SAME ->
    conv2d 1x1, as ONE launch of the wide kernel): z bit-identical to ds_maxpool_fwd / ds_maxpool_bn_relu_fwd followed by the
    plain launch wherever that also runs on the wide kernel (same MFMA sequence per output element), the recorded winners
    identical byte for byte -- on inputs with many exact ties (values on a coarse grid, half of them ReLU zeros), so the
    row-major-first rule is exercised -- the statistics equal up to the grouping of the partial sums, and everything within
    2e-4 of the fp64 oracle."""
    ops = _ops()
    N, H, W, K, Nc, norm, bit = case
    rng = np.random.RandomState(7 + H + K)
    x = np.round(rng.normal(size=(N, H, W, K)) * 3) / 3              # coarse grid: exact ties inside most windows
    r = (np.abs(rng.normal(size=K)) + 0.5).astype(np.float32)
    sh = (rng.normal(size=K) * 0.3).astype(np.float32)
    if norm:
        r[:K // 4], sh[:K // 4] = 1.0, 0.0                           # a Branch_0 slice: activations already
        x[..., :K // 4] = np.maximum(x[..., :K // 4], 0)
    else:
        x = np.maximum(x, 0)                                          # a materialised concat / stage pool output
    xt, rt, st = dev(x), dev(r), dev(sh)
    w = dev(rng.normal(size=(K, Nc)) * 0.1)
    pivot = dev(rng.normal(size=Nc) * 0.1)
    M = N * H * W
    # the two-launch form
    pooled = torch.empty(N, H, W, K, device="cuda")
    am_ref = torch.empty(N, H, W, K, dtype=torch.uint8, device="cuda")
    if norm:
        ops.maxpool_bn_relu_fwd(xt, rt, st, pooled, am_ref, N, H, W, K, 3, 1)
    else:
        ops.maxpool_fwd(xt, pooled, am_ref, N, H, W, K, 3, 1, "SAME")
    beta = torch.zeros(Nc, device="cuda")

    def run(fused):
        plan = ops.LayerPlan(ops.DS_CONV_FWD, ops.DS_ARITH_F32, 0, N, H, W, K, Nc, 1, 1, K, Nc, ops.DS_EPI_STATS)
        am = torch.full((N, H, W, K), 77, dtype=torch.uint8, device="cuda")
        if fused:
            assert plan.enable_pool3(am)
            if norm:
                plan.d.norm_rstd, plan.d.norm_shift = rt.data_ptr(), st.data_ptr()
        z = torch.full((M + 3, Nc), 5.0, device="cuda")             # (three guard rows behind the output)
        stats = torch.zeros(2 * Nc * plan.partials, device="cuda")
        plan.run(ops._p(xt if fused else pooled), ops._p(w), ops._p(z), stats=ops._p(stats), pivot=ops._p(pivot))
        mean, rstd, shift = (torch.empty(Nc, device="cuda") for _ in range(3))
        ops.bn_finalize(stats, plan.partials, M, Nc, beta, 1e-3, 0.9997, mean, rstd, shift, None, None, pivot=pivot)
        torch.cuda.synchronize()
        assert float((z[M:] - 5.0).abs().max()) == 0.0
        return z[:M], am, mean, rstd

    z0, _, mean0, rstd0 = run(False)
    z1, am1, mean1, rstd1 = run(True)
    assert torch.equal(am1, am_ref), "winners differ at %d of %d positions" % (int((am1 != am_ref).sum()), am_ref.numel())
    if bit:
        assert torch.equal(z0, z1)
    y = np.maximum(x * r + sh, 0) if norm else x
    ref = S.max_pool(y, 3, 1, "SAME").reshape(M, K) @ w.cpu().numpy().astype(np.float64)
    close(z1, ref)
    close(z0, ref)
    assert float((mean0 - mean1).abs().max()) <= 1e-5 * max(1.0, float(mean0.abs().max()))
    if M >= 64:          # (a single pixel has variance 0: rstd = eps^-1/2 amplifies the last bit of the two groupings)
        assert float((rstd0 - rstd1).abs().max()) <= 1e-5 * float(rstd0.abs().max())
    close(mean1, ref.mean(0), 1e-4)


@pytest.mark.parametrize("case", [(20000, 1, 1, 192, 176, False), (24, 28, 28, 192, 32, True), (64, 14, 14, 480, 192, False),
                                  (96, 14, 14, 528, 128, True), (130, 14, 14, 64, 64, False)])
def test_batch_norm_finalize_inside_the_conv_launch_is_bit_identical(case):
    """ds_conv_desc.fin / ds_conv_io.fin (ds_bn_finalize_in_launch): the last workgroup of a column tile to publish its
    statistics partials finalizes that tile's columns inside the launch -- same summation tree as ds_bn_finalize, so mean /
    rstd / shift and the moving averages have the bits of the separate launch; the tickets are left at zero (three launches
    in a row, the pivot aliasing `mean` as in the engine); also through the pooling loader (Branch_3)."""
    ops = _ops()
    from tumblr_emotions_amd import _lib
    N, H, W, K, Nc, pool = case
    rng = np.random.RandomState(3 + K)
    M = N * H * W
    xt = dev(np.maximum(rng.normal(size=(N, H, W, K)), 0))
    w = dev(rng.normal(size=(K, Nc)) * 0.1)
    beta = dev(rng.normal(size=Nc) * 0.1)
    res = []
    for fused in (False, True):
        plan = ops.LayerPlan(ops.DS_CONV_FWD, ops.DS_ARITH_F32, 0, N, H, W, K, Nc, 1, 1, K, Nc, ops.DS_EPI_STATS)
        am = torch.empty(M, K, dtype=torch.uint8, device="cuda")
        if pool:
            assert plan.enable_pool3(am)
        nt = plan.finalize_tickets()
        assert nt > 0 and plan.partials <= 256
        z = torch.empty(M, Nc, device="cuda")
        stats = torch.zeros(2 * Nc * plan.partials, device="cuda")
        mean = dev(rng.normal(size=Nc) * 0.0 + 0.05)                  # (the pivot of the first launch)
        rstd, shift = torch.empty(Nc, device="cuda"), torch.empty(Nc, device="cuda")
        mm, mv = torch.zeros(Nc, device="cuda"), torch.ones(Nc, device="cuda")
        tickets = torch.zeros(nt, dtype=torch.int32, device="cuda")
        f = _lib.BnFinalizeInLaunch()
        f.beta, f.mean, f.rstd, f.shift = beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), shift.data_ptr()
        f.moving_mean, f.moving_var, f.ticket, f.count, f.eps, f.decay = mm.data_ptr(), mv.data_ptr(), tickets.data_ptr(), M, 1e-3, 0.9997
        for _ in range(3):
            if fused:
                plan.run(ops._p(xt), ops._p(w), ops._p(z), stats=ops._p(stats), pivot=ops._p(mean), fin=C.addressof(f))
            else:
                plan.run(ops._p(xt), ops._p(w), ops._p(z), stats=ops._p(stats), pivot=ops._p(mean))
                ops.bn_finalize(stats, plan.partials, M, Nc, beta, 1e-3, 0.9997, mean, rstd, shift, mm, mv, pivot=mean)
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0
        res.append((z, mean, rstd, shift, mm, mv))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    zr = res[1][0].double().cpu().numpy()
    close(res[1][1], zr.mean(0), 1e-5)
    close(res[1][2], 1.0 / np.sqrt(zr.var(0) + 1e-3), 1e-5)


@pytest.mark.parametrize("case", [(2, 17, 24), (3, 12, 64), (1, 30, 192)])
def test_batch_norm_backward_from_the_pooled_gradient(case):
    """ds_bn_pool_bwd_reduce/_apply (BN+ReLU backward of a conv behind a 3x3/2 SAME pool, straight from the pool's
    output gradient) against the oracle's MaxPoolGrad followed by the oracle's BatchNorm backward."""
    ops = _ops()
    N, H, Cc = case
    rng = np.random.RandomState(17)
    z = rng.normal(0.2, 1.5, size=(N, H, H, Cc))
    beta = rng.normal(size=Cc) * 0.3
    M = N * H * H
    y_bn, mean, var, xhat, rstd = S.batch_norm_train(z.reshape(M, Cc), beta)
    y = np.maximum(y_bn, 0).reshape(N, H, H, Cc)
    pooled = S.max_pool(y, 3, 2, "SAME")
    OH = pooled.shape[1]
    dpool = rng.normal(size=pooled.shape)
    dy_full = S.max_pool_bwd(y, dpool, 3, 2, "SAME").reshape(M, Cc)
    g = dy_full * (y.reshape(M, Cc) > 0)
    dz_ref, dbeta_ref = S.batch_norm_train_bwd(g, xhat, rstd)
    # forward through the fused kernel to get the arg-max the backward kernels consume
    zd = dev(z)
    shift = beta - mean * rstd
    rstd_d, shift_d, mean_d = dev(rstd), dev(shift), dev(mean)
    yp = torch.empty(N, OH, OH, Cc, device="cuda")
    am = torch.empty(N, OH, OH, Cc, dtype=torch.uint8, device="cuda")
    ops.maxpool_bn_relu_fwd(zd, rstd_d, shift_d, yp, am, N, H, H, Cc, 3, 2)
    P = ops.bn_pool_bwd_partials(N, H, H, Cc)
    part = torch.zeros(2 * Cc * P, device="cuda")
    dpd = dev(dpool)
    ops.bn_pool_bwd_reduce(zd, dpd, am, N, H, H, Cc, mean_d, rstd_d, shift_d, part)
    dbeta = torch.empty(Cc, device="cuda")
    coef = torch.empty(2, Cc, device="cuda")
    ops.bn_bwd_finalize(part, P, M, Cc, dbeta, coef)
    dz = torch.empty(M, Cc, device="cuda")
    ops.bn_pool_bwd_apply(zd, dpd, am, N, H, H, Cc, mean_d, rstd_d, shift_d, coef, dz)
    torch.cuda.synchronize()
    close(yp, pooled)
    close(dbeta, dbeta_ref, 1e-4)
    close(dz, dz_ref, 2e-4)
    again = torch.zeros_like(part)
    ops.bn_pool_bwd_reduce(zd, dpd, am, N, H, H, Cc, mean_d, rstd_d, shift_d, again)
    torch.cuda.synchronize()
    assert torch.equal(part, again)                      # fixed summation order


def test_avgpool_dropout():
    ops = _ops()
    rng = np.random.RandomState(9)
    N, HW, Cc = 5, 49, 1024
    x = rng.normal(size=(N, HW, Cc))
    mask = (rng.uniform(size=(N, Cc)) < 0.8).astype(np.float64)
    ref = x.mean(1) * mask / 0.8
    out, mo = torch.empty(N, Cc, device="cuda"), torch.empty(N, Cc, device="cuda")
    ops.avgpool_dropout_fwd(dev(x), N, HW, Cc, 0.8, 0, dev(mask), mo, out)
    close(out, ref)
    close(mo, mask, 0)
    # generated mask: Bernoulli(0.8), reproducible per seed, different across seeds
    m1, m2, m3 = (torch.empty(N, Cc, device="cuda") for _ in range(3))
    ops.avgpool_dropout_fwd(dev(x), N, HW, Cc, 0.8, 123, None, m1, out)
    ops.avgpool_dropout_fwd(dev(x), N, HW, Cc, 0.8, 123, None, m2, out)
    ops.avgpool_dropout_fwd(dev(x), N, HW, Cc, 0.8, 124, None, m3, out)
    assert torch.equal(m1, m2) and not torch.equal(m1, m3)
    assert abs(float(m1.mean()) - 0.8) < 0.03
    close(out, x.mean(1) * m3.cpu().numpy() / 0.8)        # `out` holds the seed-124 call
    d = rng.normal(size=(N, Cc))
    dx = torch.empty(N, HW, Cc, device="cuda")
    ops.avgpool_dropout_bwd(dev(d), dev(mask), N, HW, Cc, 0.8, dx)
    torch.cuda.synchronize()
    close(dx, np.broadcast_to((d * mask / 0.8 / HW)[:, None, :], (N, HW, Cc)))


@pytest.mark.parametrize("D", [300, 50, 7])
def test_gather_rows_is_bit_exact(D):
    ops = _ops()
    rng = np.random.RandomState(10)
    B, T, V = 6, 9, 100
    table = rng.normal(size=(V + 1, D)).astype(np.float32)
    table[V] = 0
    ids = rng.randint(0, V + 1, size=(B, T)).astype(np.int64)
    td = dev(table)
    out = torch.empty(T, B, D, device="cuda")
    ops.gather_rows(td, dev(ids, torch.int64), out, B, T, D, True)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), table[ids].transpose(1, 0, 2))      # bit exact
    out2 = torch.empty(B, T, D, device="cuda")
    ops.gather_rows(td, dev(ids, torch.int64), out2, B, T, D, False)
    torch.cuda.synchronize()
    assert np.array_equal(out2.cpu().numpy(), table[ids])


def test_gather_rows_long_list_is_bit_exact():
    """2^17 positions out of a 10 001 x 300 table: Zipf-like ids, out-of-range ids (zero rows), a ragged last wave,
    both output orders."""
    ops = _ops()
    rng = np.random.RandomState(16)
    B, T, V, D = 1031, 127, 10000, 300
    table = rng.normal(size=(V + 1, D)).astype(np.float32)
    table[V] = 0
    ids = np.minimum(rng.zipf(1.3, size=(B, T)) - 1, V).astype(np.int64)
    ids[5, 7], ids[900, 3] = V + 5, -2                       # outside the table: zero rows
    want = np.where(((ids >= 0) & (ids <= V))[..., None], table[np.clip(ids, 0, V)], 0).astype(np.float32)
    td, idd = dev(table), dev(ids, torch.int64)
    out = torch.full((T, B, D), float("nan"), device="cuda")
    ops.gather_rows(td, idd, out, B, T, D, True)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want.transpose(1, 0, 2))
    out2 = torch.full((B, T, D), float("nan"), device="cuda")
    ops.gather_rows(td, idd, out2, B, T, D, False)
    torch.cuda.synchronize()
    assert np.array_equal(out2.cpu().numpy(), want)


@pytest.mark.parametrize("D,B,T,V", [(300, 6, 9, 40), (50, 5, 70, 30), (7, 3, 5, 200), (512, 2, 3, 4)])
def test_embedding_grad_scatter_add(D, B, T, V):
    """dtable[v] = sum of the dx rows whose id is v (np.add.at), collisions, unused rows (zeros), the pad row,
    both row orders, and bit-reproducibility (fixed summation order, no atomics)."""
    ops = _ops()
    rng = np.random.RandomState(15)
    ids = rng.randint(0, V + 1, size=(B, T)).astype(np.int64)
    ids[0, :3] = 1                                                  # guaranteed collisions
    dx = rng.normal(size=(B, T, D)).astype(np.float32)
    want = np.zeros((V + 1, D), np.float64)
    np.add.at(want, ids.reshape(-1), dx.reshape(-1, D).astype(np.float64))
    idd = dev(ids, torch.int64)
    got_tm = torch.full((V + 1, D), 7.0, device="cuda")
    dx_tm = dev(np.ascontiguousarray(dx.transpose(1, 0, 2)))        # time-major rows t*B+b
    ops.embedding_grad(dx_tm, idd, got_tm, B, T, D, True)
    got_bm = torch.full((V + 1, D), 7.0, device="cuda")
    dx_bm = dev(dx)
    ops.embedding_grad(dx_bm, idd, got_bm, B, T, D, False)
    again = torch.empty(V + 1, D, device="cuda")
    ops.embedding_grad(dx_tm, idd, again, B, T, D, True)
    torch.cuda.synchronize()
    close(got_tm, want, 1e-5)
    assert torch.equal(got_tm, got_bm) and torch.equal(got_tm, again)
    unused = np.setdiff1d(np.arange(V + 1), ids.reshape(-1))
    assert unused.size == 0 or float(got_tm[torch.as_tensor(unused, device="cuda")].abs().max()) == 0.0


def test_lstm_cell_forward_backward():
    ops = _ops()
    rng = np.random.RandomState(11)
    B, H, t = 7, 24, 3
    pre = rng.normal(size=(B, 4 * H))
    c0, h0 = rng.normal(size=(B, H)), rng.normal(size=(B, H))
    seq = np.array([5, 1, 4, 3, 9, 2, 4])
    i, j, f, o = np.split(pre, 4, axis=1)
    si, tj, sf, so = S.sigmoid(i), np.tanh(j), S.sigmoid(f + 1.0), S.sigmoid(o)
    live = (t < seq)[:, None]
    cn = c0 * sf + si * tj
    hn = np.tanh(cn) * so
    g = dev(pre)
    cd, hd = torch.empty(B, H, device="cuda"), torch.empty(B, H, device="cuda")
    sd = dev(seq, torch.int64)
    ops.lstm_cell_fwd(g, dev(c0), dev(h0), sd, t, B, H, 1.0, cd, hd)
    close(cd, np.where(live, cn, c0), 1e-5)
    close(hd, np.where(live, hn, h0), 1e-5)
    close(g, np.concatenate([si, tj, sf, so], 1), 1e-5)
    dh, dc = rng.normal(size=(B, H)), rng.normal(size=(B, H))
    tc = np.tanh(cn)
    dct = dc + dh * so * (1 - tc ** 2)
    ref_dg = np.where(live, np.concatenate([dct * tj * si * (1 - si), dct * si * (1 - tj ** 2),
                                            dct * c0 * sf * (1 - sf), dh * tc * so * (1 - so)], 1), 0)
    dg = torch.empty(B, 4 * H, device="cuda")
    dcp, dhc = torch.empty(B, H, device="cuda"), torch.empty(B, H, device="cuda")
    ops.lstm_cell_bwd(g, cd, dev(c0), dev(dh), dev(dc), sd, t, B, H, dg, dcp, dhc)
    torch.cuda.synchronize()
    close(dg, ref_dg, 1e-5)
    close(dcp, np.where(live, dct * sf, dc), 1e-5)
    close(dhc, np.where(live, 0, dh), 1e-6)


def test_softmax_ce_and_adam_and_reductions():
    ops = _ops()
    rng = np.random.RandomState(12)
    B, Cc = 300, 15
    z = rng.normal(size=(B, Cc)) * 3
    y = rng.randint(0, Cc, size=B)
    loss, dl = torch.empty(1, device="cuda"), torch.empty(B, Cc, device="cuda")
    up = torch.full((1,), 0.5, device="cuda")
    ops.softmax_ce(dev(z), dev(y, torch.int64), B, Cc, 1.0, up, loss, dl)
    close(loss, np.array([S.softmax_cross_entropy(z, y)]), 1e-5)
    close(dl, 0.5 * S.softmax_cross_entropy_grad(z, y), 1e-5)
    # TF Adam, three steps, weight decay on the first 40 entries, averaged gradients (grad_scale)
    n, nwd = 1003, 40
    w = rng.normal(size=n)
    m, v = np.zeros(n), np.zeros(n)
    wd_, md, vd = dev(w), dev(m), dev(v)
    for t in range(1, 4):
        gr = rng.normal(size=n) * 1e-3
        ge = gr * 0.5
        ge[:nwd] += S.WEIGHT_DECAY * w[:nwd]
        w, m, v = S.adam_step(w, ge, m, v, t, 1e-3)
        lr_t = 1e-3 * np.sqrt(1 - S.ADAM_B2 ** t) / (1 - S.ADAM_B1 ** t)
        ops.adam_tf(wd_, dev(gr), md, vd, n, nwd, S.WEIGHT_DECAY, 0.5, lr_t, S.ADAM_B1, S.ADAM_B2, S.ADAM_EPS)
    close(wd_, w, 1e-5)
    x = rng.normal(size=(777, 130))
    scratch, out = torch.empty(64 * 130, device="cuda"), torch.empty(130, device="cuda")
    ops.colsum(dev(x), 777, 130, 130, scratch, out)
    close(out, x.sum(0), 1e-5)
    for rows in (256, 3, 512):           # few row splits: the one-launch form (head biases)
        ops.colsum(dev(x[:rows]), rows, 130, 130, scratch, out)
        close(out, x[:rows].sum(0), 1e-5)
    ss = torch.empty(1, device="cuda")
    ops.sumsq(dev(x), x.size, scratch, ss)
    torch.cuda.synchronize()
    close(ss, np.array([(x ** 2).sum()]), 1e-5)


def test_every_conv_instantiation_matches_the_oracle(tuning_lib):
    """All tile shapes of the three kernel families (register-staged LDS: mt 1-2 x nt 1-6; register-direct:
    mt 1-2 x nt 1-4; LDS-DMA with the 32-deep K-tile: nt 1-3), forward (n-contiguous weights, BN statistics) and dgrad (k-contiguous, flipped taps), on a
    shape with ragged M, a K tail and a partial last column tile -- reached through the tuning knobs."""
    ops = _ops()
    from tumblr_emotions_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(14)
    N, H, W, Ci, Co, k = 3, 13, 11, 40, 200, 3
    x = rng.normal(size=(N, H, W, Ci))
    w = rng.normal(size=(k, k, Ci, Co)) * 0.1
    dy = rng.normal(size=(N, H, W, Co))
    fwd_ref = S.conv2d_same(x, w, 1).reshape(-1, Co)
    dgr_ref = S.conv2d_same_bwd_input(dy, w, (N, H, W, Ci), 1).reshape(-1, Ci)
    xd, wd, dyd = dev(x), dev(w), dev(dy)
    try:
        for path, nts in ((1, range(1, 7)), (2, range(1, 5)), (3, range(1, 4))):
            for mt in (1, 2):
                for nt in nts:
                    assert lib.ds_debug_conv_set_path(path) == 0 and lib.ds_debug_conv_set_tile(mt, nt) == 0
                    plan = ops.ConvPlan(N, H, W, Ci, Ci, k, k, 1, Co, Co, Ci * Co, 1, Co, flags=ops.DS_EPI_STATS)
                    z = torch.empty(plan.M, Co, device="cuda")
                    stats = torch.zeros(2, Co, plan.partials, device="cuda")
                    plan.run(ops._p(xd), ops._p(wd), ops._p(z), stats=ops._p(stats))
                    g = ops.ConvPlan(N, H, W, Co, Co, k, k, 1, Ci, Ci, Ci * Co, Co, 1, flip=1)
                    dx = torch.empty(g.M, Ci, device="cuda")
                    g.run(ops._p(dyd), ops._p(wd), ops._p(dx))
                    torch.cuda.synchronize()
                    tag = "path %d tile %d,%d" % (path, mt, nt)
                    assert np.abs(z.cpu().numpy() - fwd_ref).max() <= 2e-4 * np.abs(fwd_ref).max(), tag
                    assert np.abs(stats[0].sum(1).cpu().numpy() - fwd_ref.sum(0)).max() <= 1e-3 * np.abs(fwd_ref.sum(0)).max() + 1e-3, tag
                    assert np.abs(dx.cpu().numpy() - dgr_ref).max() <= 2e-4 * np.abs(dgr_ref).max(), tag
    finally:
        lib.ds_debug_conv_set_path(0)
        lib.ds_debug_conv_set_tile(0, 0)


def test_split_k_gemm_slabs_feed_lstm_cell():
    """LSTM step as the engine runs it: split-K recurrent GEMM -> slabs -> cell kernel adds them."""
    ops = _ops()
    rng = np.random.RandomState(13)
    B, H, S = 40, 96, 3
    h0, c0 = rng.normal(size=(B, H)), rng.normal(size=(B, H))
    wh = rng.normal(size=(H, 4 * H)) * 0.2
    xw = rng.normal(size=(B, 4 * H))
    seq = rng.randint(1, 6, size=B)
    t = 2
    slabs = torch.full((S, B, 4 * H), 7.0, device="cuda")
    hd, whd = dev(h0), dev(wh)
    ops.gemm_plan(B, H, 4 * H, H, 4 * H, 4 * H, splits=S, z_split_stride=B * 4 * H).run(
        ops._p(hd), ops._p(whd), ops._p(slabs))
    torch.cuda.synchronize()
    close(slabs.sum(0), h0 @ wh, 1e-5)
    g = dev(xw)
    cd, hn = torch.empty(B, H, device="cuda"), torch.empty(B, H, device="cuda")
    ops.lstm_cell_fwd(g, dev(c0), hd, dev(seq, torch.int64), t, B, H, 1.0, cd, hn, slabs, S, B * 4 * H)
    pre = xw + h0 @ wh
    i, j, f, o = np.split(pre, 4, axis=1)
    cn = c0 * S_sigmoid(f + 1.0) + S_sigmoid(i) * np.tanh(j)
    live = (t < seq)[:, None]
    torch.cuda.synchronize()
    close(cd, np.where(live, cn, c0), 1e-5)
    close(hn, np.where(live, np.tanh(cn) * S_sigmoid(o), h0), 1e-5)


def S_sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


_LSTM_PLACEMENT_SCRIPT = r"""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, %r)
from tumblr_emotions_amd import ops
out = []
for (B, T, H) in ((48, 7, 128), (64, 9, 512), (256, 5, 512)):
    rng = np.random.RandomState(B + H)
    pre = torch.from_numpy((rng.normal(size=(T, B, 4 * H)) * 0.7).astype(np.float32)).cuda()
    wh = torch.from_numpy((rng.normal(size=(H, 4 * H)) * (0.5 / np.sqrt(H))).astype(np.float32)).cuda()
    seq = torch.from_numpy(rng.randint(1, T + 1, size=B).astype(np.int64)).cuda()
    dh = torch.from_numpy(rng.normal(size=(B, H)).astype(np.float32)).cuda()
    h, c = torch.zeros(T + 1, B, H, device="cuda"), torch.zeros(T + 1, B, H, device="cuda")
    dg = torch.empty(T, B, 4 * H, device="cuda")
    ws = torch.zeros(max(ops.lstm_seq_workspace(B, H) // 4, 4), dtype=torch.int32, device="cuda")
    for rep in range(2):          # the second pass runs on a used workspace / ring
        g = pre.clone()
        ops.lstm_seq_fwd(g, ops._p(wh), 4 * H, h, c, seq, T, B, H, 1.0, ws)
        ops.lstm_seq_bwd(g, ops._p(wh), 4 * H, c, dh, H, seq, T, B, H, dg, ws)
        torch.cuda.synchronize()
        ops.lstm_seq_status(ws, B)
    for t in (h, c, g, dg):
        out.append(hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest())
print("DIGEST " + " ".join(out))
"""


def test_lstm_exchange_is_placement_independent():
    """The persistent LSTM picks its hand-off form from the MEASURED placement (row group on one XCD: plain stores into the
    fragment-order ring; spread over XCDs: write-through).  Both forms -- the XCD-local 1-D launch and the plain 2-D grid
    (DS_LSTM_XCD=0, which spreads a row group over all XCDs) -- must give the same bits, for the 16-row (B = 48, 64) and the
    32-row (B = 256) groups, also on a re-used workspace."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for xcd in ("1", "0"):
        # (the knob exists in the tuning build only: the subprocess loads libds_kernels_tuning.so)
        env = dict(os.environ, DS_LSTM_XCD=xcd, DS_LIB=os.path.join(root, "tumblr_emotions_amd", "libds_kernels_tuning.so"))
        r = subprocess.run([sys.executable, "-c", _LSTM_PLACEMENT_SCRIPT % root], env=env, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST ")]
        assert line, r.stdout[-2000:]
        digests.append(line[-1])
    assert digests[0] == digests[1]


@pytest.mark.parametrize("case", [(256, 1024, 256, False), (32, 512, 15, False), (256, 512, 512, True), (64, 768, 512, False),
                                  (300, 256, 40, True)])
def test_split_gemm_with_slab_epilogue_matches_the_single_launch(case):
    """head_gemm_plan: split-K ds_conv_igemm + ds_slab_epilogue (bias / accumulate / mask / relu in ds_conv_igemm's order) for the
    batch x features GEMMs of the heads -- against fp64 and against the single launch; strided input and output rows; two
    runs give the same bits (slabs combined in index order)."""
    ops = _ops()
    M, K, N, transposed = case
    rng = np.random.RandomState(M + K + N)
    a = rng.normal(size=(M, K))
    w = rng.normal(size=(N, K) if transposed else (K, N)) * 0.05
    bias, prev = rng.normal(size=N), rng.normal(size=(M, N))
    mask = (rng.uniform(size=(M, N)) < 0.6).astype(np.float64)
    ad = dev(np.pad(a, ((0, 0), (0, 8))))                       # row stride K + 8
    wd, biasd, maskd = dev(w), dev(bias), dev(np.pad(mask, ((0, 0), (0, 4))))
    core = a @ (w.T if transposed else w)
    for flags, want in ((ops.DS_EPI_BIAS | ops.DS_EPI_RELU, np.maximum(core + bias, 0)),
                        (ops.DS_EPI_ACCUM | ops.DS_EPI_BIAS | ops.DS_EPI_RELU, np.maximum(core + bias + prev, 0)),
                        (ops.DS_EPI_MASK, core * mask), (0, core)):
        plan = ops.head_gemm_plan(M, K, N, K + 8, N + 4, K if transposed else N, transposed_w=transposed, flags=flags,
                                  ldmask=N + 4, device="cuda")
        assert isinstance(plan, ops.SplitGemm)
        outs = []
        for rep in range(2):
            out = dev(np.pad(prev, ((0, 0), (0, 4)), constant_values=5.0))
            plan.run(ops._p(ad), ops._p(wd), ops._p(out), bias=ops._p(biasd), mask=ops._p(maskd))
            torch.cuda.synchronize()
            outs.append(out)
        assert torch.equal(outs[0], outs[1])
        close(outs[0][:, :N], want, 2e-4)
        assert bool((outs[0][:, N:] == 5.0).all())              # the row padding is untouched
        single = ops.gemm_plan(M, K, N, K + 8, N + 4, K if transposed else N, transposed_w=transposed, flags=flags, ldmask=N + 4)
        ref = dev(np.pad(prev, ((0, 0), (0, 4)), constant_values=5.0))
        single.run(ops._p(ad), ops._p(wd), ops._p(ref), bias=ops._p(biasd), mask=ops._p(maskd))
        torch.cuda.synchronize()
        assert float((outs[0] - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    # below the thresholds the plain plan is returned
    assert not isinstance(ops.head_gemm_plan(4096, 512, 64, 512, 64, 64, device="cuda"), ops.SplitGemm)
    assert not isinstance(ops.head_gemm_plan(64, 15, 512, 15, 512, 15, transposed_w=True, device="cuda"), ops.SplitGemm)
