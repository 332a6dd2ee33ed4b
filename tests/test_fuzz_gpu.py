"""Randomised parity sweep of the conv entry points (forward with every epilogue, dgrad, wgrad) against the NumPy
oracle in fp64: kernel sizes 1/3/5, stride 1/2, channel counts that are not multiples of 4 (scalar-load variants),
Cout = 15, every kernel family and tile width reachable through the tuning knobs.  Fixed seed; the former scratch/fuzz_conv.py
runs the same loop with more cases (400 cases, 0 mismatches on MI355X this round)."""
import numpy as np
import pytest
import torch

from oracle import tf_semantics as S

pytestmark = pytest.mark.gpu


def test_random_conv_shapes_forward_dgrad_wgrad(tuning_lib):
    from tumblr_emotions_amd import _lib, ops
    lib = _lib.load()
    rng = np.random.RandomState(2026)
    ncases = 60
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    failures = []
    try:
        for case in range(ncases):
            k = int(rng.choice([1, 1, 3, 3, 5]))
            stride = int(rng.choice([1, 1, 1, 2]))
            N = int(rng.randint(1, 5)); H = int(rng.randint(max(k, 2), 20)); W = int(rng.randint(max(k, 2), 20))
            Ci = int(rng.choice([4, 8, 12, 16, 20, 24, 32, 36, 48, 64, 96, 100, 3, 7])); Co = int(rng.choice([4, 8, 15, 16, 32, 40, 64, 96, 100, 160, 200]))
            path = int(rng.choice([0, 0, 1, 3])); nt = int(rng.choice([0, 0, 1, 2, 3]))
            x = rng.normal(size=(N, H, W, Ci)); w = rng.normal(size=(k, k, Ci, Co)) * 0.2
            ref = S.conv2d_same(x, w, stride)
            OH, OW = ref.shape[1], ref.shape[2]
            lib.ds_debug_conv_set_path(path); lib.ds_debug_conv_set_tile(1 if nt else 0, nt)
            flags = int(rng.choice([0, ops.DS_EPI_STATS, ops.DS_EPI_BIAS | ops.DS_EPI_RELU, ops.DS_EPI_ACCUM]))
            bias = rng.normal(size=Co); prev = rng.normal(size=(N * OH * OW, Co))
            plan = ops.ConvPlan(N, H, W, Ci, Ci, k, k, stride, Co, Co, Ci * Co, 1, Co, flags=flags)
            z = dev(prev) if flags & ops.DS_EPI_ACCUM else torch.empty(plan.M, Co, device="cuda")
            stats = torch.zeros(2, Co, max(plan.partials, 1), device="cuda")
            xd, wd, bd = dev(x), dev(w), dev(bias)
            plan.run(ops._p(xd), ops._p(wd), ops._p(z), bias=ops._p(bd), stats=ops._p(stats))
            want = ref.reshape(-1, Co)
            if flags & ops.DS_EPI_BIAS: want = np.maximum(want + bias, 0)
            if flags & ops.DS_EPI_ACCUM: want = want + prev
            torch.cuda.synchronize()
            got = z.cpu().numpy().astype(np.float64)
            tol = 3e-4 * max(1.0, np.abs(want).max())
            ok = np.abs(got - want).max() <= tol
            if flags & ops.DS_EPI_STATS:
                ok = ok and np.abs(stats[0].sum(1).cpu().numpy() - want.sum(0)).max() <= 2e-3 * max(1.0, np.abs(want.sum(0)).max())
                ok = ok and np.abs(stats[1].sum(1).cpu().numpy() - (want ** 2).sum(0)).max() <= 2e-3 * max(1.0, (want ** 2).sum(0).max())
            # dgrad (stride 1 only) and wgrad of the same geometry
            if stride == 1:
                dy = rng.normal(size=(N, OH, OW, Co))
                g = ops.ConvPlan(N, H, W, Co, Co, k, k, 1, Ci, Ci, Ci * Co, Co, 1, flip=1)
                dx = torch.empty(g.M, Ci, device="cuda"); dyd = dev(dy)
                g.run(ops._p(dyd), ops._p(wd), ops._p(dx))
                dref = S.conv2d_same_bwd_input(dy, w, (N, H, W, Ci), 1).reshape(-1, Ci)
                torch.cuda.synchronize()
                ok = ok and np.abs(dx.cpu().numpy() - dref).max() <= 3e-4 * max(1.0, np.abs(dref).max())
                wp = ops.WgradPlan(N, H, W, Ci, Ci, k, k, 1, Co, Co)
                ws = torch.empty(max(wp.ws_bytes // 4, 1), device="cuda"); dw = torch.empty(k, k, Ci, Co, device="cuda")
                wp.run(ops._p(xd), ops._p(dyd), ops._p(dw), ops._p(ws), wp.ws_bytes)
                wref = S.conv2d_same_bwd_filter(x, dy, (k, k, Ci, Co), 1)
                torch.cuda.synchronize()
                ok = ok and np.abs(dw.cpu().numpy() - wref).max() <= 5e-4 * max(1.0, np.abs(wref).max())
            if not ok:
                failures.append("case %d: N=%d H=%d W=%d Ci=%d Co=%d k=%d stride=%d path=%d nt=%d flags=%d"
                                % (case, N, H, W, Ci, Co, k, stride, path, nt, flags))
    finally:
        lib.ds_debug_conv_set_path(0)
        lib.ds_debug_conv_set_tile(0, 0)
    assert not failures, "conv parity mismatches:\n" + "\n".join(failures)


def test_random_shapes_round2_kernels(tuning_lib):
    """The same sweep for the kernels added in round 2, each forced on regardless of the selection rules: fused Winograd
    (forward with statistics and dgrad; odd extents, ragged channel blocks, strided rows), the wide register-direct 1x1
    kernel (every NB, both weight orientations, accumulate epilogue), the packed-RGB stem, the register-direct wgrad
    (M >= 512: every vector width, stride 2, ragged slabs) and the register-direct bf16 conv (against the oracle on
    bf16-rounded operands)."""
    from tumblr_emotions_amd import _lib, ops
    lib = _lib.load()
    rng = np.random.RandomState(4052)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")

    def bf16r(a):
        u = np.asarray(a, np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32).astype(np.float64)

    def bad(got, want, tol):
        return np.abs(got.detach().cpu().numpy().astype(np.float64).reshape(want.shape) - want).max() > tol * max(1.0, np.abs(want).max())

    failures = []
    try:
        for case in range(24):                                   # ---- Winograd
            N = int(rng.randint(1, 5)); H = int(rng.randint(3, 23)); W = int(rng.randint(3, 23))
            Ci = int(rng.choice([8, 16, 24, 40, 64, 96])); Co = int(rng.choice([8, 16, 31, 32, 48, 72, 100, 160]))
            pad = int(rng.choice([0, 0, 4]))
            x = rng.normal(size=(N, H, W, Ci + pad)); w = rng.normal(size=(3, 3, Ci, Co)) * 0.2
            xs = x[..., :Ci]
            ref = S.conv2d_same(xs, w, 1).reshape(-1, Co)
            xd, wd = dev(x), dev(w)
            pv = dev(rng.normal(size=Co))
            dy = rng.normal(size=(N, H, W, Co)); dyd = dev(dy)
            um = ref - pv.double().cpu().numpy()
            for f4 in (False, True):                             # F(2x2,3x3) and, where the shape allows, F(4x4,3x3)
                ok = True
                if not f4 or ops.wino4_supported(H, W, Ci, Co):
                    plan = ops.WinoPlan(N, H, W, Ci, Ci + pad, Co, Co, flags=ops.DS_EPI_STATS, f4=f4)
                    u = torch.empty(plan.u_elems, device="cuda")
                    ops.wino_transform_weights(ops._p(wd), u, Ci, Co, False, f4=f4)
                    z = torch.full((plan.M, Co), float("nan"), device="cuda")
                    stats = torch.zeros(2, Co, plan.partials, device="cuda")
                    plan.run(ops._p(xd), ops._p(u), ops._p(z), stats=ops._p(stats), pivot=ops._p(pv))
                    torch.cuda.synchronize()
                    ok = not bad(z, ref, 3e-4) and not bad(stats[0].sum(1), um.sum(0), 2e-3) and not bad(stats[1].sum(1), (um ** 2).sum(0), 2e-3)
                if Co % 8 == 0 and (not f4 or ops.wino4_supported(H, W, Co, Ci)):
                    g = ops.WinoPlan(N, H, W, Co, Co, Ci, Ci, f4=f4)
                    ug = torch.empty(g.u_elems, device="cuda")
                    ops.wino_transform_weights(ops._p(wd), ug, Ci, Co, True, f4=f4)
                    dx = torch.full((g.M, Ci), float("nan"), device="cuda")
                    g.run(ops._p(dyd), ops._p(ug), ops._p(dx))
                    torch.cuda.synchronize()
                    ok = ok and not bad(dx, S.conv2d_same_bwd_input(dy, w, (N, H, W, Ci), 1).reshape(-1, Ci), 3e-4)
                if not ok:
                    failures.append("winograd %s case %d: N=%d H=%d W=%d Ci=%d Co=%d pad=%d" % ("F4x4" if f4 else "F2x2", case, N, H, W, Ci, Co, pad))
        lib.ds_debug_conv_set_wide(2)
        for case in range(24):                                   # ---- wide 1x1
            M = int(rng.randint(1, 700)); K = int(rng.choice([32, 40, 64, 72, 104, 192, 296])); Nn = int(rng.choice([8, 24, 32, 40, 64, 96, 104, 160, 200, 224, 256, 300]))
            a = rng.normal(size=(M, K)); w = rng.normal(size=(K, Nn)) * 0.2; prev = rng.normal(size=(M, Nn))
            ad, wd = dev(a), dev(w)
            out = dev(prev)
            plan = ops.ConvPlan(M, 1, 1, K, K, 1, 1, 1, Nn, Nn, 0, 1, Nn, flags=ops.DS_EPI_ACCUM, pad_t=0, pad_l=0, OH=1, OW=1)
            plan.run(ops._p(ad), ops._p(wd), ops._p(out))
            torch.cuda.synchronize()
            ok = not bad(out, prev + a @ w, 3e-4)
            if Nn % 8 == 0 and Nn >= 32:
                dz = rng.normal(size=(M, Nn)); dzd = dev(dz)
                g = ops.gemm_plan(M, Nn, K, Nn, K, Nn, transposed_w=True)
                dx = torch.full((M, K), float("nan"), device="cuda")
                g.run(ops._p(dzd), ops._p(wd), ops._p(dx))
                torch.cuda.synchronize()
                ok = ok and not bad(dx, dz @ w.T, 3e-4)
            if not ok:
                failures.append("wide case %d: M=%d K=%d N=%d" % (case, M, K, Nn))
        lib.ds_debug_conv_set_wide(1)
        for case in range(8):                                    # ---- stem
            N = int(rng.randint(1, 4)); H = int(rng.randint(7, 70)); W = int(rng.randint(7, 70)); cs = int(rng.choice([3, 4]))
            x = rng.uniform(-1, 1, size=(N, H, W, 3)); w = rng.normal(size=(7, 7, 3, 64)) * 0.1
            ref = S.conv2d_same(x, w, 2).reshape(-1, 64)
            wst = np.zeros((7, 7, cs, 64)); wst[:, :, :3] = w
            plan = ops.StemPlan(N, H, W, cs, 64, 64)
            z = torch.full((plan.M, 64), float("nan"), device="cuda")
            xd, wd = dev(x), dev(wst)
            plan.run(ops._p(xd), ops._p(wd), ops._p(z))
            torch.cuda.synchronize()
            if bad(z, ref, 3e-4):
                failures.append("stem case %d: N=%d H=%d W=%d cs=%d" % (case, N, H, W, cs))
        for case in range(16):                                   # ---- register-direct wgrad
            k = int(rng.choice([1, 3])); stride = int(rng.choice([1, 1, 2]))
            N = int(rng.randint(4, 12)); H = int(rng.randint(9, 20)); W = int(rng.randint(9, 20))
            Ci = int(rng.choice([5, 8, 12, 16, 34, 48, 100])); Co = int(rng.choice([6, 15, 16, 36, 64, 100]))
            x = rng.normal(size=(N, H, W, Ci))
            OH, OW = -(-H // stride), -(-W // stride)
            if N * OH * OW < 512:
                continue
            dy = rng.normal(size=(N, OH, OW, Co))
            wp = ops.WgradPlan(N, H, W, Ci, Ci, k, k, stride, Co, Co)
            ws = torch.empty(max(wp.ws_bytes // 4, 1), device="cuda"); dw = torch.full((k, k, Ci, Co), float("nan"), device="cuda")
            xd, dyd = dev(x), dev(dy)
            wp.run(ops._p(xd), ops._p(dyd), ops._p(dw), ops._p(ws), wp.ws_bytes)
            torch.cuda.synchronize()
            if bad(dw, S.conv2d_same_bwd_filter(x, dy, (k, k, Ci, Co), stride), 5e-4):
                failures.append("wgrad case %d: N=%d H=%d W=%d Ci=%d Co=%d k=%d stride=%d" % (case, N, H, W, Ci, Co, k, stride))
        for case in range(16):                                   # ---- register-direct bf16
            k = int(rng.choice([1, 3])); stride = int(rng.choice([1, 1, 2]))
            N = int(rng.randint(1, 4)); H = int(rng.randint(3, 18)); W = int(rng.randint(3, 18))
            Ci = int(rng.choice([8, 16, 24, 40, 72])); Co = int(rng.choice([8, 24, 32, 56, 100, 200, 264]))
            x = rng.normal(size=(N, H, W, Ci)); w = rng.normal(size=(k, k, Ci, Co)) * 0.2
            ref = S.conv2d_same(bf16r(x), bf16r(w), stride)
            xd, wd = dev(x), dev(w)
            wb = torch.empty(ops.weights_bf16_bytes(Ci, Co, k * k, 0), dtype=torch.uint8, device="cuda")
            ops.weights_to_bf16(ops._p(wd), wb, Ci, Co, k * k, 0)
            plan = ops.Bf16Plan(N, H, W, Ci, Ci, k, stride, Co, Co)
            z = torch.full((plan.M, Co), float("nan"), device="cuda")
            plan.run(ops._p(xd), ops._p(wb), ops._p(z))
            torch.cuda.synchronize()
            if bad(z, ref.reshape(-1, Co), 3e-4):
                failures.append("bf16 case %d: N=%d H=%d W=%d Ci=%d Co=%d k=%d stride=%d" % (case, N, H, W, Ci, Co, k, stride))
    finally:
        lib.ds_debug_conv_set_wide(1)
    assert not failures, "parity mismatches:\n" + "\n".join(failures)


def test_fuzz_harness_reports_a_planted_mismatch():
    """The sweep above must be able to fail: the same comparison with a deliberately wrong reference
    (one weight perturbed) has to be flagged."""
    from tumblr_emotions_amd import ops
    rng = np.random.RandomState(7)
    N, H, W, Ci, Co, k = 2, 9, 9, 16, 32, 3
    x = rng.normal(size=(N, H, W, Ci)); w = rng.normal(size=(k, k, Ci, Co)) * 0.2
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    plan = ops.ConvPlan(N, H, W, Ci, Ci, k, k, 1, Co, Co, Ci * Co, 1, Co)
    z = torch.empty(plan.M, Co, device="cuda")
    xd, wd = dev(x), dev(w)                      # keep the operands alive until the launch has run
    plan.run(ops._p(xd), ops._p(wd), ops._p(z))
    torch.cuda.synchronize()
    got = z.cpu().numpy().astype(np.float64)
    good = S.conv2d_same(x, w, 1).reshape(-1, Co)
    w_bad = w.copy(); w_bad[1, 1, 3, 5] += 0.05
    bad = S.conv2d_same(x, w_bad, 1).reshape(-1, Co)
    tol = 3e-4 * max(1.0, np.abs(good).max())
    assert np.abs(got - good).max() <= tol
    assert np.abs(got - bad).max() > tol
