"""Test helper: the ReLU / max-pool decisions the HIP image tower took in its last forward pass, in the
format oracle.torch_ref.DeepSentimentRef.inject understands.  Reads engine buffers only (activations are
kept for the backward pass); never part of the product path."""
import numpy as np

from oracle import tf_semantics as S


def _np(t):
    """activation buffer -> numpy (bf16 under 16-bit activation storage: widened exactly)"""
    return t.float().cpu().numpy()


def keep_activations(net):
    """Call BEFORE the step whose decisions are wanted.  By default the 3x3 / Branch_3 convs of Mixed_3b..4e leave
    pre-BatchNorm values in the concat buffers (zcat) and the backward pass differentiates them in place, so their
    ReLU masks cannot be read back afterwards; this switches the engine to the materialised form, which
    test_zcat_step_is_bit_identical proves to give the same bits."""
    if net.image is not None and net.image.zcat:
        net.image.zcat, net.image.B = False, None


def hip_decisions(net):
    from tumblr_emotions_amd.engine_image import ConvStage, MixedStage, PoolStage
    eng = net.image
    assert not any(getattr(st, "zcat", False) for st in eng.stages), "call keep_activations(net) before the step"
    inj = {}
    for st in eng.stages:
        if isinstance(st, ConvStage):
            scope = "InceptionV1/" + st.name
            if st.fused_into_pool:       # BatchNorm + ReLU run behind the pool: relu(rstd*max(z)+shift)
                inj["norelu/" + scope] = True
            else:
                inj["relu/" + scope] = (st.out > 0).cpu().numpy()
        elif isinstance(st, PoolStage):
            if getattr(st, "zmax", None) is not None:
                # MaxPool_2a ran INSIDE the stem kernel (ds_conv_stem_pool): neither the full-resolution z nor the winners
                # exist.  Rebuild both with the two-launch form from the same images, weights and batch statistics -- z has the
                # same bits (same MFMA sequence; test_stem_with_the_pool_inside_...), which the pooled maxima confirm here
                import torch
                from tumblr_emotions_amd import ops
                lay = st.prev.layer
                plan = ops.LayerPlan(ops.DS_CONV_FWD, eng.arith, eng.plan_options() | ops.DS_PLAN_PACKED_RGB, lay.B, lay.H, lay.W,
                                     4, lay.cout, 7, 2, 4, lay.cout, 0)
                z_full = torch.empty(lay.M, lay.cout, device=eng.device)
                plan.run(ops._p(eng.images), lay.w_ptr, ops._p(z_full))
                y = torch.empty(lay.B, st.H, st.W, st.C, device=eng.device)
                am = torch.empty(lay.B, st.H, st.W, st.C, dtype=torch.uint8, device=eng.device)
                ops.maxpool_bn_relu_fwd(z_full, lay.rstd, lay.shift, y, am, lay.B, lay.OH, lay.OW, st.C, 3, 2)
                torch.cuda.synchronize()
                want = torch.clamp_min(torch.addcmul(lay.shift, st.zmax, lay.rstd), 0)
                assert float((y - want).abs().max()) <= 1e-6 * max(1.0, float(y.abs().max()))
                inj["pool/" + st.name] = am.cpu().numpy()
                inj["poolrelu/" + st.name] = (y > 0).cpu().numpy()
            elif getattr(st.prev, "fused_into_pool", False):
                inj["pool/" + st.name] = st.argmax.cpu().numpy()
                inj["poolrelu/" + st.name] = (st.out > 0).cpu().numpy()
            else:
                inj["pool/" + st.name] = S.max_pool_argmax(_np(st.prev.out), st.k, st.stride, "SAME")
        elif isinstance(st, MixedStage):
            b0, b1a, b1b, b2a, b2b, b3 = st.b
            B, H, W = st.B, st.H, st.W
            out = _np(st.out)
            pre = "InceptionV1/%s/" % st.name
            nm = [n for (n, _, _, _) in S.mixed_conv_names(st.name)]
            o1, o2, o3 = b0, b0 + b1b, b0 + b1b + b2b
            inj["relu/" + pre + nm[0]] = out[..., :o1] > 0
            inj["relu/" + pre + nm[1]] = _np(st.r1.view(B, H, W, b1a)) > 0
            inj["relu/" + pre + nm[2]] = out[..., o1:o2] > 0
            inj["relu/" + pre + nm[3]] = _np(st.r2.view(B, H, W, b2a)) > 0
            inj["relu/" + pre + nm[4]] = out[..., o2:o3] > 0
            inj["relu/" + pre + nm[5]] = out[..., o3:] > 0
            inj["pool/%s/Branch_3" % st.name] = S.max_pool_argmax(_np(st.prev.out), 3, 1, "SAME")
    assert sum(k.startswith(("relu/", "norelu/")) for k in inj) == 57
    return inj
