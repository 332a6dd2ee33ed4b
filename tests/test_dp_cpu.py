"""Data-parallel path on CPU: world_size 2 over gloo (the RCCL path runs the same host code).

Checks (slim/deployment/model_deploy.py:221-223,301-302,414-444):
  * the two-bucket sum-all-reduce over the flat gradient buffer and the 1/world scale;
  * rank r owns the r-th contiguous slice of the same seeded global batch;
  * DP-equivalence on a model without BatchNorm (text tower): averaging the per-rank gradients of the
    per-rank mean losses equals the gradient of the global-batch mean loss.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tf_semantics as S
from oracle import torch_ref as R
from tumblr_emotions_amd.dp import GradientReducer, shard_bounds
from tumblr_emotions_amd.params import ParamStore


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _text_problem():
    rng = np.random.RandomState(31)
    V, D, H, T, B = 30, 8, 12, 7, 6
    params = R.make_params("text", rng, num_classes=15, embed_dim=D, rnn_size=H, dtype=np.float64)
    emb = S.synthetic_embedding(V, D).astype(np.float64)
    batch = S.synthetic_batch(B, T, V, seed=2, with_images=False)
    return params, emb, batch


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    from tumblr_emotions_amd import dp
    assert dp.init_distributed("gloo", rank=rank, world_size=world) == (rank, world)
    try:
        with pytest.raises(RuntimeError):          # a second call would reserve the streams too late: refused
            dp.init_distributed("gloo", rank=rank, world_size=world)
        # ---- bucketed all-reduce over a flat buffer laid out by ParamStore --------------------------
        st = ParamStore("cpu")
        st.declare("conv/weights", (3, 5), True, l2=True, bucket=1)
        st.declare("head/W", (7,), True, bucket=1)
        st.declare("upstream/beta", (6,), True, bucket=2)
        st.finalize()
        assert st.n_l2 == 16 and st.n_bucket1 == 24 and st.n_trainable_padded == 32
        st.grad.copy_(torch.arange(32, dtype=torch.float32) * (rank + 1))
        red = GradientReducer(st.grad, st.n_bucket1, overlap=True)
        assert red.world == world and not red.overlap          # no side stream on CPU tensors
        red.expect("a", "b")
        red.begin_step()
        red.stage_done("a")
        red.stage_done("b")
        scale = red.finish()
        assert scale == 0.5
        torch.testing.assert_close(st.grad, torch.arange(32, dtype=torch.float32) * 3)

        # ---- DP equivalence on the text tower (oracle as the compute) ----------------------------------
        params, emb, batch = _text_problem()
        n = batch["labels"].shape[0]
        lo, hi = shard_bounds(n, rank, world)
        local = {k: v[lo:hi] for k, v in batch.items()}
        ref = R.DeepSentimentRef(params, emb, "text", torch.float64)
        g_local = ref.train_step(local, 1e-3)["grads"]
        names = sorted(g_local)
        flat = torch.cat([g_local[k].reshape(-1) for k in names])
        red2 = GradientReducer(flat, flat.numel() // 2)
        red2.begin_step()
        flat.mul_(red2.finish())                               # what the Adam kernel does with grad_scale
        if rank == 0:
            full = R.DeepSentimentRef(params, emb, "text", torch.float64).train_step(batch, 1e-3)["grads"]
            ref_flat = torch.cat([full[k].reshape(-1) for k in names])
            out.put(float((flat - ref_flat).abs().max() / ref_flat.abs().max()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_bucketed_allreduce_and_dp_equivalence():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, "rank failed (exit code %r)" % p.exitcode
    assert out.get(timeout=10) < 1e-12


def test_shard_bounds_cover_the_batch_exactly_once():
    for n in (256, 7, 64):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_param_store_layout_and_tf_names_roundtrip():
    st = ParamStore("cpu")
    st.declare("InceptionV1/Mixed_5c/fused_1x1/weights", (1, 1, 4, 6), True, l2=True, bucket=1,
               columns=[("InceptionV1/Mixed_5c/Branch_0/Conv2d_0a_1x1/weights", 0, 2),
                        ("InceptionV1/Mixed_5c/Branch_1/Conv2d_0a_1x1/weights", 2, 5),
                        ("InceptionV1/Mixed_5c/Branch_2/Conv2d_0a_1x1/weights", 5, 6)])
    st.declare("frozen/weights", (2, 3), False)
    st.declare("b", (3,), True, bucket=2)
    st.finalize()
    rng = np.random.RandomState(0)
    sd = {n: rng.normal(size=(1, 1, 4, c1 - c0)) for (n, c0, c1) in st.entries[
        "InceptionV1/Mixed_5c/fused_1x1/weights"].columns}
    sd["frozen/weights"] = rng.normal(size=(2, 3))
    sd["b"] = rng.normal(size=3)
    st.load_state_dict(sd)
    back = st.state_dict()
    assert set(back) == set(sd)
    for k in sd:
        np.testing.assert_allclose(back[k], sd[k].astype(np.float32), rtol=0, atol=0)
    # fused tensor really is the column concatenation, 16-byte aligned offsets, L2 prefix first
    fused = st.view("InceptionV1/Mixed_5c/fused_1x1/weights").numpy()
    np.testing.assert_allclose(fused[..., 2:5], sd["InceptionV1/Mixed_5c/Branch_1/Conv2d_0a_1x1/weights"].astype(np.float32))
    assert st.entries["InceptionV1/Mixed_5c/fused_1x1/weights"].offset == 0 and st.n_l2 == 24
    assert all(e.offset % 4 == 0 for e in st.entries.values())


def test_bench_started_without_a_launcher_relaunches_itself_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (how the driver starts a bench) must become the
    N-rank torch.distributed.run job instead of failing: the re-exec argv is the contract's launcher line, rendezvous on
    127.0.0.1, the caller's arguments unchanged; under a launcher (WORLD_SIZE set) nothing is re-executed and a rank
    count that disagrees with --gpus is an error, not an assert."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ds_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = bench.self_launch_argv(8, ["--gpus", "8", "--steps", "3", "--warmup", "1"], port=29999)
    assert argv == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                    "--master-addr", "127.0.0.1", "--master-port", "29999", os.path.join(root, "bench.py"),
                    "--gpus", "8", "--steps", "3", "--warmup", "1"]
    port = int(bench.self_launch_argv(2, [])[9])
    assert 1024 < port < 65536
    # the real thing, end to end on CPU: a stand-in for `python` records what bench.py exec's
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    src = open(os.path.join(root, "bench.py")).read()
    probe = tmp_path / "bench_probe.py"
    probe.write_text(src.replace("os.execv(cmd[0], cmd)", "print('EXEC ' + ' '.join(cmd[1:])); sys.exit(0)"))
    r = subprocess.run([sys.executable, str(probe), "--gpus", "4", "--steps", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("EXEC -m torch.distributed.run --nnodes=1 --nproc-per-node 4 "), r.stderr
    assert r.stdout.rstrip().endswith("--gpus 4 --steps 2")
