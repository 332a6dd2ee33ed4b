"""Data-parallel step with the real kernels: two ranks sharing the one GPU of the test box, gloo as the
transport (RCCL refuses two ranks on one device; the host code path -- sharding, early bucket-1 launch
on a side stream, event ordering across the text-tower stream, 1/world scale inside Adam -- is the
same one `bench.py --gpus N` runs over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

TEXT = dict(nb_emotions=15, rnn_size=32, vocab_size=50, embedding_dim=20, post_size=10)


def _collect(out, procs, n, timeout=600):
    """n results from the queue; a worker that died (exception on the GPU box) fails the test at once instead of
    after the queue timeout."""
    import queue
    import time
    got, t0 = [], time.time()
    while len(got) < n:
        try:
            got.append(out.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, "a worker exited with %s" % dead
            assert time.time() - t0 < timeout, "timeout waiting for the workers"
    return got


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from tumblr_emotions_amd import dp
    dp.init_distributed("gloo", device=0, rank=rank, world_size=world)
    try:
        from tumblr_emotions_amd.net import SentimentNet
        from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
        res = {}
        for mode in ("text", "joint"):
            net = SentimentNet(mode=mode, **TEXT)
            assert net.world == world and net.reducer.overlap
            net.initialize(seed=3)
            gb = synthetic_batch_numpy(8, 10, 50, seed=4, with_images=(mode == "joint"))
            local = to_device(gb, "cuda", rank, world)
            assert local["labels"].shape[0] == 4
            for _ in range(2):
                net.train_step(local, 1e-3, seed=11)
            torch.cuda.synchronize()
            res[mode] = net.store.theta.detach().cpu().numpy()
        out.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_steps_on_one_gpu():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(_collect(out, procs, 2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # both ranks applied the same reduced gradient: bit-identical parameters
    for mode in ("text", "joint"):
        assert np.array_equal(got[0][mode], got[1][mode]), mode
        assert np.isfinite(got[0][mode]).all()

    # the BatchNorm-free text model is DP-invariant: 2 ranks x 4 samples == 1 process x 8 samples
    from tumblr_emotions_amd.net import SentimentNet
    from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
    net = SentimentNet(mode="text", **TEXT)
    net.initialize(seed=3)
    batch = to_device(synthetic_batch_numpy(8, 10, 50, seed=4, with_images=False))
    for _ in range(2):
        net.train_step(batch, 1e-3, seed=11)
    torch.cuda.synchronize()
    single = net.store.theta.detach().cpu().numpy()
    # Adam's sign-like first steps amplify last-bit differences of tiny gradients to ~lr: compare loosely
    # everywhere and tightly in the mean
    assert np.abs(single - got[0]["text"]).max() <= 2.5e-3
    assert np.abs(single - got[0]["text"]).mean() <= 2e-5


def _rccl_world1_worker(port, out):
    """One rank, backend nccl (= RCCL on ROCm): the bucketed path with the early bucket-1 launch on the side stream,
    RCCL's async work handle and the event ordering across the text-tower stream -- the code path of an 8-GPU run."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from tumblr_emotions_amd import dp
    dp.init_distributed("nccl", device=0, rank=0, world_size=1)
    try:
        from tumblr_emotions_amd.net import SentimentNet
        from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
        probe = torch.ones(1, device="cuda")
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        res = {"rccl_ranks": int(round(float(probe.item()))), "backend": dist.get_backend()}
        batch = to_device(synthetic_batch_numpy(4, 10, 50, seed=4, with_images=True))
        for forced in (False, True):
            net = SentimentNet(mode="joint", force_dp_buckets=forced, **TEXT)
            assert net.reducer.active == forced and net.reducer.overlap == forced
            net.initialize(seed=3)
            if forced:
                net.reducer.timing = True
            for _ in range(3):
                net.train_step(batch, 1e-3, seed=11)
            torch.cuda.synchronize()
            res["theta_%d" % forced] = net.store.theta.detach().cpu().numpy()
            res["grad_%d" % forced] = net.store.grad.detach().cpu().numpy()
            if forced:
                res["report"] = net.reducer.overlap_report()
                assert net.capture_step(batch) is False        # RCCL stays outside a captured graph
        out.put(res)
    finally:
        dist.destroy_process_group()


def test_rccl_single_rank_bucketed_allreduce_is_the_identity():
    """RCCL itself (backend "nccl"), which gloo tests cannot exercise.  With one rank the sum is the
    identity and the scale is 1/1, so three training steps through the forced bucket path -- all-reduce of bucket 1
    launched from inside the backward pass on the side stream with async_op=True, bucket 2 at the end -- must leave
    bit-identical parameters and gradients to the plain step; the early reduce must really have run (event report)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(_free_port(), out))
    p.start()
    res = _collect(out, [p], 1)[0]
    p.join(60)
    assert p.exitcode == 0
    assert res["rccl_ranks"] == 1 and res["backend"] == "nccl"
    print("rccl_ranks == %d (backend %s); bucket-1 report: %s" % (res["rccl_ranks"], res["backend"], res["report"]))
    assert np.array_equal(res["theta_0"], res["theta_1"])
    assert np.array_equal(res["grad_0"], res["grad_1"])
    rep = res["report"]
    assert rep and rep["bucket1_bytes"] > 0 and rep["bucket1_allreduce_ms"] > 0.0


# ---- DP parity of the joint (BatchNorm) model as SURVEY 8(e) defines it -------------------------------------------------
JOINT = dict(nb_emotions=15, im_features_size=256, rnn_size=32, fc_size=512, vocab_size=60, embedding_dim=20, post_size=12)
DP_SEED, DP_PER_RANK, DP_LR = 71, 8, 1e-3


def _dp_parity_inputs():
    from oracle import tf_semantics as S
    from oracle import torch_ref as R
    rng = np.random.RandomState(DP_SEED)
    params = R.make_params("joint", rng, num_classes=15, im_features_size=256, embed_dim=JOINT["embedding_dim"],
                           rnn_size=JOINT["rnn_size"], fc_size=512, dtype=np.float64)
    for k in params:
        if k.endswith("beta"):
            params[k] = rng.normal(0, 0.1, size=params[k].shape)
    emb = S.synthetic_embedding(JOINT["vocab_size"], JOINT["embedding_dim"]).astype(np.float64)
    batch = S.synthetic_batch(2 * DP_PER_RANK, JOINT["post_size"], JOINT["vocab_size"], seed=DP_SEED + 1)
    return params, emb, batch


def _dp_parity_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from tumblr_emotions_amd import dp
    dp.init_distributed("gloo", device=0, rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from hip_decisions import hip_decisions, keep_activations
        from tumblr_emotions_amd.net import SentimentNet
        params, emb, batch = _dp_parity_inputs()
        net = SentimentNet(mode="joint", dropout_keep_prob=1.0, **JOINT)
        assert net.world == world and net.reducer.active
        net.load_state_dict(dict(params, **{"Text/W_embedding": emb}))
        lo, hi = rank * DP_PER_RANK, (rank + 1) * DP_PER_RANK
        local = {k: torch.from_numpy(np.ascontiguousarray(v[lo:hi])).cuda() for k, v in batch.items()}
        keep_activations(net)
        net.train_step(local, DP_LR)
        torch.cuda.synchronize()
        res = dict(logits=net.logits.detach().cpu().numpy(), loss=net.total_loss_value(), decisions=hip_decisions(net),
                   grads=net.grads_state_dict(), after=net.state_dict(),
                   l2=[e.name for e in net.store.entries.values() if e.trainable and e.l2])
        out.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_two_rank_joint_step_matches_the_clone_oracle():
    """SURVEY 8(e) / VERDICT r03 #7: the data-parallel JOINT step (BatchNorm inside) against the oracle run the way slim
    defines clones (slim/deployment/model_deploy.py:221-223,301-302,353-355,414-444; oracle: DeepSentimentRef.train_step_dp):
    every rank normalises with the batch statistics of ITS sub-batch, the two cross-entropy gradients are averaged, the L2
    term is counted once, rank 0's moving statistics are the clone-0 update.  Two ranks on the one GPU (gloo carries the
    all-reduce; bucketing, early bucket-1 launch, 1/world and L2 inside the Adam kernel are the RCCL run's code).  Gates as
    in test_model_gpu._check_step: logits and loss 1e-3 per rank, every one of the 71 gradients 1e-3 (relative L2 and
    max-norm) along the HIP ranks' own ReLU / pool decisions, the un-injected oracle within 1e-4 of the injected one, TF-Adam
    1e-5 on resolved entries, moving statistics 1e-5."""
    from oracle import torch_ref as R
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_parity_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(_collect(out, procs, 2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    params, emb, batch = _dp_parity_inputs()
    subs = [{k: v[r * DP_PER_RANK:(r + 1) * DP_PER_RANK] for k, v in batch.items()} for r in range(2)]
    ref = R.DeepSentimentRef(params, emb, "joint", torch.float64)
    with torch.no_grad():
        plain = [ref.forward(s).detach().clone() for s in subs]
    o = ref.train_step_dp(subs, DP_LR, injects=[got[0]["decisions"], got[1]["decisions"]])
    reg = o["loss"] - 0.5 * (o["ce"][0] + o["ce"][1])
    for r in range(2):
        assert float((o["logits"][r] - plain[r]).abs().max()) <= 1e-4
        assert np.abs(got[r]["logits"] - o["logits"][r].numpy()).max() <= 1e-3, r
        assert abs(got[r]["loss"] - (o["ce"][r] + reg)) <= 1e-3, r          # rank r reports ITS cross-entropy + the L2 term
    # the reduced gradient is the same buffer on both ranks; what Adam consumes is sum / world + L2 (applied in the kernel)
    assert len(o["grads"]) == 71
    worst = (0.0, "")
    for name, g_ref in o["grads"].items():
        g_ref = g_ref.numpy()
        assert np.array_equal(got[0]["grads"][name], got[1]["grads"][name]), name
        g = got[0]["grads"][name].reshape(g_ref.shape) / 2.0
        if name in got[0]["l2"]:
            g = g + 0.00004 * params[name]
        d = g - g_ref
        rel = np.linalg.norm(d) / max(np.linalg.norm(g_ref), 1e-30)
        emax = np.abs(d).max() / max(np.abs(g_ref).max(), 1e-30)
        worst = max(worst, (rel, name))
        assert rel <= 1e-3 and emax <= 1e-3, "gradient of %s: relative L2 %.3e, max-norm %.3e" % (name, rel, emax)
    print("DP joint step, worst gradient relative L2 against the clone oracle: %.3e (%s)" % worst)
    for name in ref.trainable:
        w_ref = ref.p[name].detach().numpy()
        assert np.array_equal(got[0]["after"][name], got[1]["after"][name]), name
        g_ref = o["grads"][name].numpy()
        big = np.abs(g_ref) > 1e-2 * max(np.abs(g_ref).max(), 1e-12)
        assert (np.abs(got[0]["after"][name].reshape(w_ref.shape) - w_ref)[big] <= 1e-5).mean() >= 0.99, name
    for name, v in got[0]["after"].items():
        if name.endswith("moving_mean") or name.endswith("moving_variance"):
            np.testing.assert_allclose(v, ref.p[name].numpy(), atol=1e-5, err_msg=name)
    # per-rank statistics: rank 1 averaged ITS sub-batch, so its moving means differ from rank 0's
    k = "InceptionV1/Conv2d_2b_1x1/BatchNorm/moving_mean"
    assert not np.array_equal(got[0]["after"][k], got[1]["after"][k])


def _sync_bn_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from tumblr_emotions_amd import dp
    dp.init_distributed("gloo", device=0, rank=rank, world_size=world)
    try:
        from tumblr_emotions_amd.net import SentimentNet
        from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
        net = SentimentNet(mode="joint", dropout_keep_prob=1.0, sync_bn=True, **TEXT)
        assert net.sync_bn and net.image.sync_world == world
        net.initialize(seed=3)
        local = to_device(synthetic_batch_numpy(8, 10, 50, seed=4, with_images=True), "cuda", rank, world)
        net.train_step(local, 1e-3)
        torch.cuda.synchronize()
        res = dict(logits=net.logits.detach().cpu().numpy(), ce=float(net.loss_buf.item()), grads=net.grads_state_dict(),
                   after=net.state_dict())
        # second run: BOTH ranks get the same four samples.  Doubling every partial sum and the count is exact in floating
        # point, so the synchronised step must give the bits of one process on those four samples -- any partial-sum
        # region left out of an all-reduce, or a count not multiplied by the world size, breaks that
        net2 = SentimentNet(mode="joint", dropout_keep_prob=1.0, sync_bn=True, **TEXT)
        net2.initialize(seed=5)
        same = to_device(synthetic_batch_numpy(4, 10, 50, seed=6, with_images=True))
        for _ in range(2):
            net2.train_step(same, 1e-3)
        torch.cuda.synchronize()
        res["theta_same"] = net2.store.theta.detach().cpu().numpy()
        res["frozen_same"] = net2.store.frozen.detach().cpu().numpy()
        out.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_sync_bn_reproduces_the_single_process_step():
    """SURVEY 8(e), optional `sync_bn=True`: BatchNorm statistics and the two column means of its backward formula are
    all-reduced per layer, so two ranks x 4 samples compute what one process computes on the 8 samples (slim's default --
    and this build's -- is per-clone statistics; see test_two_rank_joint_step_matches_the_clone_oracle).  Same function,
    different summation order: logits 1e-4, cross-entropy 1e-5, moving statistics 1e-5, the head gradients that follow the
    logits 1e-4 / 1e-3; every other gradient to 3e-2 (median 2e-2): below a ReLU two fp32 evaluations with different summation
    orders differ by ~1e-2 at any batch size (profiles/r02_oracle_fp32_spread.txt), which is what is seen here."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_bn_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(_collect(out, procs, 2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from tumblr_emotions_amd.net import SentimentNet
    from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device
    net = SentimentNet(mode="joint", dropout_keep_prob=1.0, **TEXT)
    net.initialize(seed=3)
    net.train_step(to_device(synthetic_batch_numpy(8, 10, 50, seed=4, with_images=True)), 1e-3)
    torch.cuda.synchronize()
    logits = net.logits.detach().cpu().numpy()
    both = np.concatenate([got[0]["logits"], got[1]["logits"]])
    assert np.abs(both - logits).max() <= 1e-4, np.abs(both - logits).max()
    assert abs(0.5 * (got[0]["ce"] + got[1]["ce"]) - float(net.loss_buf.item())) <= 1e-5
    single = net.grads_state_dict()
    rels, by_name = [], {}
    for name, g in single.items():
        assert np.array_equal(got[0]["grads"][name], got[1]["grads"][name]), name
        d = got[0]["grads"][name] / 2.0 - g                      # reduced sum / world
        rels.append(np.linalg.norm(d) / max(np.linalg.norm(g), 1e-30))
        by_name[name] = rels[-1]
        assert rels[-1] <= 3e-2, (name, rels[-1])
    print(sorted(by_name.items(), key=lambda kv: kv[1])[:8])
    # what sits above every ReLU of the image tower follows the logits: tight
    assert by_name["b_softmax"] <= 1e-4 and by_name["W_softmax"] <= 1e-3, (by_name["b_softmax"], by_name["W_softmax"])
    assert np.median(rels) <= 2e-2, np.median(rels)
    after = net.state_dict()
    for name, v in after.items():
        if name.endswith("moving_mean") or name.endswith("moving_variance"):
            np.testing.assert_allclose(got[0]["after"][name], v, atol=1e-5, err_msg=name)
            assert np.array_equal(got[0]["after"][name], got[1]["after"][name]), name
    # identical shards on both ranks: bit-identical to one process on that shard (see the worker)
    net2 = SentimentNet(mode="joint", dropout_keep_prob=1.0, **TEXT)
    net2.initialize(seed=5)
    same = to_device(synthetic_batch_numpy(4, 10, 50, seed=6, with_images=True))
    for _ in range(2):
        net2.train_step(same, 1e-3)
    torch.cuda.synchronize()
    for r in range(2):
        assert np.array_equal(got[r]["theta_same"], net2.store.theta.detach().cpu().numpy())
        assert np.array_equal(got[r]["frozen_same"], net2.store.frozen.detach().cpu().numpy())      # (moving statistics)
    print("sync_bn: max|dlogits| %.2e, median gradient rel L2 %.2e, worst %.2e" % (np.abs(both - logits).max(), np.median(rels), max(rels)))


def test_bench_self_launched_two_rank_run_prints_one_json_line():
    """The N-rank bench path end to end, the way the driver starts it: plain `python bench.py --gpus 2` (no launcher in the
    environment) must re-execute itself under torch.distributed.run, bring up two ranks, time the weak-scaling headline and
    the strong-scaling point of BASELINE configs[3], and have rank 0 print exactly ONE JSON line.  On this 1-GPU box both
    ranks share cuda:0 and gloo carries the all-reduce (DS_BENCH_ONE_DEVICE=1: RCCL refuses two ranks on one device) -- the
    host code path, sharding, bucket schedule and report are the ones an 8-GPU run over RCCL takes
    (slim/deployment/model_deploy.py:414-444 is what the all-reduce replaces)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DS_BENCH_ONE_DEVICE"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "32", "--no-cpu-baseline", "--no-live-traffic", "--no-gather"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    if r.returncode != 0:
        # one retry: the launcher picks a free rendezvous port by binding and releasing it, and two ranks share one device here --
        # seen failing once in ~10 full-suite runs on a fresh box and never alone; a second failure is a real one
        print("first attempt failed (rc %d):\n%s" % (r.returncode, r.stderr[-2000:]))
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "re-executing as" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout[-4000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["dp"]["rccl_ranks"] == 2 and rec["dp"]["world_size"] == 2
    assert rec["config"]["global_batch"] == 64 and rec["config"]["per_gpu_batch"] == 32
    assert rec["value"] > 0 and abs(rec["value"] - 64 / (rec["ms_per_step"] * 1e-3)) <= 1e-2 * rec["value"]
    ss = rec["strong_scaling"]
    assert ss["scaling"] == "strong" and ss["n_gpus"] == 2 and ss["per_gpu_batch"] * 2 == ss["global_batch"] and ss["value"] > 0
    assert np.isfinite(rec["config"]["final_loss"])
