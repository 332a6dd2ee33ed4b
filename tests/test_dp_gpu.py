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
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
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
