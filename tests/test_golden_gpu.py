"""HIP path against the committed oracle regression vector tests/golden/joint_step_oracle.npz
(one joint training step, B=2; generated at build time by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_joint_step_matches_committed_golden_vector():
    from oracle import tf_semantics as S
    from oracle import torch_ref as R
    from tumblr_emotions_amd.net import SentimentNet
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "joint_step_oracle.npz"))
    cfg = json.loads(str(g["cfg"]))
    rng = np.random.RandomState(cfg["param_seed"])          # same seeded construction as make_golden.py
    params = R.make_params("joint", rng, num_classes=15, im_features_size=256, embed_dim=cfg["D"],
                           rnn_size=cfg["H"], fc_size=512, dtype=np.float64)
    emb = S.synthetic_embedding(cfg["V"], cfg["D"])
    batch = S.synthetic_batch(cfg["B"], cfg["T"], cfg["V"], seed=cfg["batch_seed"])
    net = SentimentNet(mode="joint", nb_emotions=15, im_features_size=256, rnn_size=cfg["H"], fc_size=512,
                       vocab_size=cfg["V"], embedding_dim=cfg["D"], post_size=cfg["T"])
    net.load_state_dict(dict(params, **{"Text/W_embedding": emb}))
    dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    net.train_step(dev, cfg["lr"], dropout_mask=torch.from_numpy(g["dropout_mask"]).cuda())
    torch.cuda.synchronize()
    assert np.abs(net.logits.detach().cpu().numpy() - g["logits"]).max() <= 1e-3
    assert abs(net.total_loss_value() - float(g["loss"])) <= 1e-3
    grads = net.grads_state_dict()
    for key in g.files:
        if key.startswith("grad/"):
            ref = g[key].astype(np.float64)
            got = grads[key[5:]].reshape(ref.shape)
            rel = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12)
            assert rel <= 2e-2, (key, rel)      # B=2: ReLU-mask flips move a few channels (see test_model_gpu.py)


def test_joint_step_b16_matches_committed_golden_vector():
    """HIP step against tests/golden/joint_step_b16_oracle.npz (fp64 oracle, B = 16, made by
    tests/golden/make_golden_step.py): logits / loss to 1e-3, moving statistics to 1e-5, one TF-Adam step, and
    the gradient of every one of the 71 trainable variables.  This is a PLAIN fp64 comparison (no decision
    injection), so each gradient is gated by 3x the fp32-vs-fp64 spread the oracle itself shows for that
    variable (stored in the fixture; floor: 1e-3, and the tower's median spread for variables that sit below
    ReLU / arg-max decisions, because flips are rare events)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden_step import build
    from tumblr_emotions_amd.net import SentimentNet
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "joint_step_b16_oracle.npz"))
    cfg = json.loads(str(g["cfg"]))
    params, emb, batch, mask = build(cfg)
    net = SentimentNet(mode="joint", nb_emotions=15, im_features_size=256, rnn_size=cfg["H"], fc_size=512,
                       vocab_size=cfg["V"], embedding_dim=cfg["D"], post_size=cfg["T"])
    net.load_state_dict(dict(params, **{"Text/W_embedding": emb}))
    dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    net.train_step(dev, cfg["lr"], dropout_mask=torch.tensor(mask, dtype=torch.float32).cuda())
    torch.cuda.synchronize()
    assert np.abs(net.logits.detach().cpu().numpy() - g["logits"]).max() <= 1e-3
    assert abs(net.total_loss_value() - float(g["loss"])) <= 1e-3
    grads = net.grads_state_dict()
    names = [k[5:] for k in g.files if k.startswith("grad/")]
    assert len(names) == 71 and set(names) == set(grads)
    below = [n for n in names if n.startswith("InceptionV1/") and "/Logits/" not in n]
    floor = float(np.median([float(g["spread/" + n]) for n in below]))
    report = []
    for n in names:
        ref = g["grad/" + n].astype(np.float64)
        got = grads[n].reshape(-1)
        got = got[::cfg["stride"]] if got.size > cfg["big"] else got
        rel = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
        gate = max(1e-3, 3 * max(float(g["spread/" + n]), floor if n in below else 0.0))
        report.append((rel / gate, rel, gate, n))
        assert rel <= gate, "gradient of %s: relative L2 %.3e above %.3e" % (n, rel, gate)
    report.sort(reverse=True)
    print("closest to its gate: %s rel %.3e gate %.3e; median rel %.3e"
          % (report[0][3], report[0][1], report[0][2], float(np.median([r[1] for r in report]))))
    after = net.state_dict()
    for k in g.files:
        if k.startswith("moving/"):
            np.testing.assert_allclose(after[k[7:]], g[k], atol=1e-5, err_msg=k)
        elif k.startswith("adam/"):          # one TF-Adam step moves an entry by at most ~lr; tight where resolved
            w, w_ref = after[k[5:]].reshape(-1), g[k]
            assert np.abs(w - w_ref).max() <= 2.5 * cfg["lr"] + 1e-6, k
            gr = g["grad/" + k[5:]]
            if gr.size == w_ref.size:
                big = np.abs(gr) > 1e-1 * np.abs(gr).max()
                assert (np.abs(w - w_ref)[big] <= 1e-5).mean() >= 0.97, k


@pytest.mark.parametrize("which", ["joint", "image"])
def test_full_size_step_matches_committed_golden_vector(which):
    """BASELINE configs[2] (joint, B = 256, T = 32, V = 10 000, D = 300, H = 512) and configs[1] (image-only, B = 128) at
    their REAL sizes against vectors the fp64 oracle produced in the build container
    (tests/golden/make_golden_fullsize.py: 53 s / 33 GB and 17 s / 17 GB of host work -- nothing a GPU box should
    repeat): logits and loss (CE + L2) to 1e-3, the gradients of the Logits conv, the dense heads and the LSTM (all
    above the tower's decisions: 1e-3 relative L2), and five BatchNorm beta gradients at different depths, each gated
    by 3x the fp32-vs-fp64 spread the oracle itself shows for it (floor 1e-3) because this is a plain comparison
    without decision injection."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden_fullsize as G
    from tumblr_emotions_amd.net import SentimentNet
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", G.CFGS[which]["file"]))
    cfg = json.loads(str(g["cfg"]))
    assert cfg["B"] == (256 if which == "joint" else 128) and cfg["mode"] == which
    params, emb, batch, mask = G.build(cfg)
    net = SentimentNet(mode=which, nb_emotions=15, im_features_size=256, rnn_size=cfg["H"], fc_size=512,
                       vocab_size=cfg["V"], embedding_dim=cfg["D"], post_size=cfg["T"])
    sd = dict(params)
    if emb is not None:
        sd["Text/W_embedding"] = emb
    net.load_state_dict(sd)
    del params, sd
    dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items()
           if which == "joint" or k in ("images", "labels")}
    net.train_step(dev, cfg["lr"], dropout_mask=torch.tensor(mask, dtype=torch.float32).cuda())
    torch.cuda.synchronize()
    dl = np.abs(net.logits.detach().cpu().numpy() - g["logits"]).max()
    dloss = abs(net.total_loss_value() - float(g["loss"]))
    assert dl <= 1e-3 and dloss <= 1e-3, (dl, dloss)
    grads = net.grads_state_dict()
    names = [k[5:] for k in g.files if k.startswith("grad/")]
    assert len(names) == (13 if which == "joint" else 7)
    # The dense layer's ReLU (relu(concat W_fc + b_fc): 256 x 512 decisions) sits between the logits and every gated gradient
    # except W_softmax / b_softmax.  Nineteen of its units have pre-activations within 1e-4 of zero for this batch -- the HIP
    # path's forward noise there is 4e-5 -- and ONE flipped unit moves those gradients by 2-5e-3 (profiles/r06_notes.md: a
    # different grouping of the stem's statistics partials flipped the unit at +4.4e-6).  The fixture therefore carries the
    # oracle's decisions and, per undecidable unit, the (exactly linear) change of each gradient when it flips
    # (tests/golden/make_golden_dense_flips.py): the oracle is evaluated ALONG the HIP path's dense decisions, as
    # tests/hip_decisions.py does for the tower, and a decision may differ only at a unit the oracle itself cannot resolve.
    flipped = []
    if which == "joint":
        want = np.unpackbits(g["dense_mask"])[:cfg["B"] * 512].reshape(cfg["B"], 512).astype(bool)
        got_mask = (net.head.dense.detach() > 0).cpu().numpy()
        units = [tuple(u) for u in g["flip/units"].tolist()]
        for b, j in zip(*np.nonzero(want != got_mask)):
            assert (int(b), int(j)) in units, "dense unit (%d, %d): ReLU decision differs from the oracle's, whose pre-activation " \
                "is not within 1e-4 of zero" % (b, j)
            flipped.append(units.index((int(b), int(j))))
    report = []
    for n in names:
        ref = g["grad/" + n].astype(np.float64)
        for i in flipped:
            if "flip/%d/%s" % (i, n) in g.files:
                ref = ref + g["flip/%d/%s" % (i, n)].astype(np.float64)
        got = grads[n].reshape(-1)
        got = got[::cfg["stride"]] if got.size > cfg["big"] else got
        rel = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
        gate = max(1e-3, 3 * float(g["spread/" + n]))
        report.append((rel, gate, n))
        assert rel <= gate, "gradient of %s: relative L2 %.3e above %.3e" % (n, rel, gate)
    if flipped:
        print("dense units decided the other way (oracle pre-activations %s): the oracle follows them"
              % ", ".join("%+.1e" % float(g["flip/pre"][i]) for i in flipped))
    print("%s B=%d: max|dlogits| %.2e, |dloss| %.2e; %s" % (which, cfg["B"], dl, dloss,
          "; ".join("%s %.1e/%.1e" % (n.split("/")[-3] if n.count("/") > 2 else n, r, gt) for r, gt, n in report)))


def test_joint_fp8_config5_share_vs_emulating_oracle():
    """BASELINE configs[4] ("fp16 joint with fp8 (CDNA4) MFMA conv path, batch 1024, 8 GPUs") as ONE rank sees it: the
    joint step at B = 128 and the real dims in the fp8 configuration AS IT SHIPS -- ds_conv_fp8 (e4m3 x e4m3 forward,
    per-tensor power-of-two scales) on the layers where it beats the bf16 kernels (the forward 3x3 convs with >= 96 input
    channels on 14 x 14 and larger maps), bf16 multiplies on the others, 16-bit (bf16) activation storage, fp32 accumulation / BatchNorm / weight gradients / text
    tower -- against the committed fp64 oracle vector that EMULATES those multiplies (tests/golden/make_golden_fp8.py,
    DeepSentimentRef.conv_multiply = "fp8_auto").  Not the 1e-3 parity path.  TOLERANCE of this configuration, stated
    here: logits within 0.35 and total loss within 0.05 of the emulating oracle, and CLOSER to it than to the exact-multiply
    oracle (measured on MI355X, printed below: 0.16 against 0.25; e4m3
    carries 3 mantissa bits and this randomly initialised 57-layer BatchNorm stack amplifies a forward perturbation
    ~100x -- the oracle with EXACT multiplies sits 0.25 from the emulating one itself, recorded in the fixture); the
    gradient of b_softmax within 0.05 relative L2, the other head / LSTM / Logits gradients reported (0.1-0.35).
    The fixture models the multiplies only: not the 16-bit activation storage and not the centred bf16 z storage of round 6
    (30 layers here, the fp8-forward ones included; it moved the distance to the emulating vector from 0.16 to 0.186, the gates unchanged)."""
    import sys
    GOLD = os.path.join(os.path.dirname(__file__), "golden")
    sys.path.insert(0, GOLD)
    from make_golden_fullsize import build
    from make_golden_fp8 import CFG
    from tumblr_emotions_amd import ops
    from tumblr_emotions_amd.net import SentimentNet
    G = np.load(os.path.join(GOLD, CFG["file"]))
    params, emb, batch, mask = build(CFG, np.float32)
    net = SentimentNet(mode="joint", nb_emotions=15, im_features_size=256, rnn_size=CFG["H"], fc_size=512,
                       vocab_size=CFG["V"], embedding_dim=CFG["D"], post_size=CFG["T"], dtype="fp8")
    assert net.image.act16 and not net.image.fp8_everywhere
    net.load_state_dict(dict(params, **{"Text/W_embedding": emb}))
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in batch.items()}
    net.train_step(dev, CFG["lr"], dropout_mask=torch.tensor(mask, dtype=torch.float32).cuda())
    torch.cuda.synchronize()
    n_fp8 = sum(l.fwd.family == ops.DS_FAM_FP8D for l in net.image.layers)
    n_b16 = sum(l.fwd.family == ops.DS_FAM_BF16D for l in net.image.layers)
    n_fp8d = sum(l.dgrad is not None and l.dgrad.family == ops.DS_FAM_FP8D for l in net.image.layers)
    assert 5 <= n_fp8 <= 8 and n_fp8d == 0 and n_b16 >= 25, (n_fp8, n_fp8d, n_b16)      # fp8 where it wins, bf16 elsewhere
    logits = net.logits.detach().cpu().numpy()
    assert np.isfinite(logits).all()
    d_emul = float(np.abs(logits - G["logits/fp8_auto"]).max())
    d_exact = float(np.abs(logits - G["logits/f32"]).max())
    oracle_gap = float(np.abs(G["logits/fp8_auto"] - G["logits/f32"]).max())
    dloss = abs(net.total_loss_value() - float(G["loss/fp8_auto"]))
    grads = net.grads_state_dict()
    rels = []
    for key in G.files:
        if key.startswith("grad/") and not key.endswith("/BatchNorm/beta"):     # (below the tower's ReLUs fp8 noise is O(1): not gated)
            name = key[5:]
            g = grads[name].reshape(-1)
            g = g[::CFG["stride"]] if g.size > CFG["big"] else g
            ref = G[key].reshape(-1)
            rels.append((float(np.linalg.norm(g - ref) / max(np.linalg.norm(ref), 1e-30)), name))
    rels.sort()
    print("fp8 configuration at B = 128 (%d fp8 / %d bf16 forward layers): max|dlogits| %.3f vs the emulating oracle, %.3f vs "
          "the exact one (the two oracles: %.3f apart); |dloss| %.4f; gradient relative L2 median %.3f, worst %.3f (%s)"
          % (n_fp8, n_b16, d_emul, d_exact, oracle_gap, dloss, rels[len(rels) // 2][0], rels[-1][0], rels[-1][1]))
    assert d_emul <= 0.35 and dloss <= 0.05 and d_emul < d_exact
    # gradients: b_softmax follows the logits alone (mean softmax - one-hot): tight; the others multiply features that
    # already carry the forward fp8 noise, so a PLAIN comparison (no decision injection is possible against a committed
    # vector) shows 0.2-0.6 -- reported, bounded for sanity
    by = {n: r for r, n in rels}
    assert by["b_softmax"] <= 0.05 and rels[-1][0] <= 0.6
