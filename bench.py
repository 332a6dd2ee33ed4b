#!/usr/bin/env python
"""Headline benchmark: Deep Sentiment training samples/s (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; started as plain
  `python bench.py --gpus N` -- no WORLD_SIZE in the environment -- it re-executes itself under exactly that launcher,
  `self_launch_argv`)

One step = forward + backward + (RCCL gradient all-reduce) + TF-Adam of train_deep_sentiment on a
synthetic batch resident in HBM: 224x224x3 images + 32-token posts, Inception-v1 + 300-d embedding +
LSTM-512, batch 256 per GPU (BASELINE cfg3; weak scaling under data parallelism -- a multi-rank run ALSO times the
strong-scaling point BASELINE configs[3] names, global batch 256 split over the ranks, and reports it in the same JSON line
as `strong_scaling`), fp32 arithmetic
(the precision the 1e-3 parity gate is stated in), dropout and BatchNorm in train mode, reference
freeze (conv weights below Mixed_5c frozen, every BatchNorm beta trainable).

Prints ONE JSON line (rank 0) carrying `roofline` for the dominant kernel (the implicit-GEMM fp32
MFMA conv/GEMM kernel, timed live with HIP events around every launch on its stream, in a second pass
of the same K steps so that the event records do not slow the headline pass) and
`cpu_baseline` (the PyTorch-CPU oracle timed on this box's host cores on a bounded sample; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# kernel arguments in device memory (this image's default; measured: forcing it OFF costs 0.3 ms per step at B = 256 and
# 0.35 ms at B = 32, profiles/r04_notes.md) -- pinned so that a differently configured runtime measures the same thing
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

GFLOP_PER_SAMPLE = 6.169          # fwd+bwd, reference-faithful freeze (BASELINE.md section 2 / SURVEY 8d)
PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA, same guide (never the 2:1-sparsity figure)
PEAK_FP8_MFMA_TFLOPS = 5000.0     # dense fp8 MFMA


def _physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def _one_socket_cpus():
    """One logical CPU per physical core of socket 0 (sysfs topology), or None when it cannot be read: the set a
    pinned single-socket run is confined to -- no cross-socket memory traffic, no SMT siblings."""
    base = "/sys/devices/system/cpu"
    try:
        allowed = os.sched_getaffinity(0)
        seen, cpus = set(), []
        for c in sorted(allowed):
            with open("%s/cpu%d/topology/physical_package_id" % (base, c)) as f:
                pkg = int(f.read())
            with open("%s/cpu%d/topology/core_id" % (base, c)) as f:
                core = int(f.read())
            if pkg == min_pkg(base, allowed) and core not in seen:
                seen.add(core)
                cpus.append(c)
        return cpus or None
    except (OSError, ValueError, AttributeError):
        return None


def min_pkg(base, allowed, _cache={}):
    if "v" not in _cache:
        pk = []
        for c in allowed:
            with open("%s/cpu%d/topology/physical_package_id" % (base, c)) as f:
                pk.append(int(f.read()))
        _cache["v"] = min(pk)
    return _cache["v"]


def cpu_baseline(post_size, vocab, dim, rnn, warmup=3, steps=10):
    """The oracle ("port" of the reference's TF-CPU step: TensorFlow 1.x cannot be installed here) timed on this
    box's host cores, SURVEY 8d: the same step (fwd + bwd + TF-Adam, same synthetic batch construction) for
    BASELINE configs[0] (text-only, batch 64, the reference's CPU-runnable case) and for the joint model of the
    headline.  Thread placements: 1 thread (the reference's `--cpus-per-task=1`, parallel_computing/job_array_train.sh:13),
    16 threads, and ONE SOCKET pinned (one thread per physical core of socket 0 through os.sched_setaffinity: what a
    well-run CPU job on this host would use -- the earlier all-cores legs ran 128 threads over two sockets and were slower
    than 16).  Every run reports its per-step mean and standard deviation.  `value` is the BEST joint throughput measured
    (the figure any speed-up should be quoted against); `value_at_headline_batch` the best at batch 256 itself."""
    import numpy as np
    import torch
    from oracle import tf_semantics as S
    from oracle import torch_ref as R
    phys = _physical_cores()
    prev = torch.get_num_threads()
    socket0 = _one_socket_cpus()
    try:
        prev_aff = os.sched_getaffinity(0)
    except AttributeError:
        prev_aff = None
    runs = []

    def timed(mode, batch, threads, warmup, steps, pin=None):
        if pin is not None:
            os.sched_setaffinity(0, pin)
        torch.set_num_threads(threads)
        try:
            rng = np.random.RandomState(1)
            params = R.make_params(mode, rng, num_classes=15, im_features_size=256, embed_dim=dim, rnn_size=rnn, fc_size=512)
            emb = S.synthetic_embedding(vocab, dim)
            b = S.synthetic_batch(batch, post_size, vocab, seed=0, with_images=(mode != "text"))
            ref = R.DeepSentimentRef(params, emb, mode, torch.float32)
            mask = None if mode == "text" else torch.tensor((rng.uniform(size=(batch, 1024)) < 0.8).astype(np.float32))
            for _ in range(warmup):
                ref.train_step(b, 1e-3, mask)
            ts = []
            for _ in range(steps):
                t0 = time.perf_counter()
                ref.train_step(b, 1e-3, mask)
                ts.append(time.perf_counter() - t0)
        finally:
            if pin is not None and prev_aff is not None:
                os.sched_setaffinity(0, prev_aff)
        mean = sum(ts) / len(ts)
        sd = (sum((t - mean) ** 2 for t in ts) / max(len(ts) - 1, 1)) ** 0.5
        runs.append(dict(workload="text-only (BASELINE configs[0])" if mode == "text" else "joint (headline model)",
                         batch=batch, threads=threads, placement="socket 0, one thread per core" if pin is not None else "unpinned",
                         warmup=warmup, steps=steps, samples_per_s=round(batch / mean, 3), sec_per_step=round(mean, 4),
                         sec_per_step_std=round(sd, 4)))

    try:
        timed("text", 64, 1, warmup, steps)
        timed("joint", 16, min(16, phys), 2, max(3, steps // 2))
        timed("joint", 64, min(16, phys), 1, 4)
        timed("joint", 256, min(16, phys), 1, 4)          # the HEADLINE batch itself: ~9 s per step on 2 x EPYC 9575F
        if socket0 and len(socket0) > 16:
            # one socket, one thread per core: probed at batch 16 first -- on the round-6 box 64 pinned threads gave 6-8 samples/s
            # with a standard deviation as large as the mean (a shared host) against 27-51 at 16 threads, so the long runs at
            # batch 64 / 256 (10-30 s per step there) are taken only when the probe beats the 16-thread figure
            timed("text", 64, len(socket0), 1, 3, pin=socket0)
            timed("joint", 16, len(socket0), 1, 3, pin=socket0)
            probe = runs[-1]["samples_per_s"]
            ref16 = [r for r in runs if r["batch"] == 16 and r["placement"] == "unpinned" and r["workload"].startswith("joint")][0]["samples_per_s"]
            if probe > ref16:
                timed("joint", 64, len(socket0), 1, 4, pin=socket0)
                timed("joint", 256, len(socket0), 1, 3, pin=socket0)
    finally:
        torch.set_num_threads(prev)
    joint = [r for r in runs if r["workload"].startswith("joint")]
    head = max((r for r in joint if r["batch"] == 256), key=lambda r: r["samples_per_s"])
    best = max(joint, key=lambda r: r["samples_per_s"])
    return dict(value=best["samples_per_s"], unit="samples/s", cores=best["threads"], kind="port",
                best_value=best["samples_per_s"], value_at_headline_batch=head["samples_per_s"],
                sample="joint train step (fwd+bwd+TF-Adam), fp32, PyTorch-CPU restatement of the TF1 step (reference-equivalent CPU "
                       "path: TensorFlow 1.x is not installable here).  `value` = the best joint throughput of the runs listed in "
                       "`runs` (batch %d, %d threads, %s: %d timed steps of %.2f +- %.2f s after %d warm-up); "
                       "`value_at_headline_batch` = the best at the headline's own batch 256 (%d threads, %s)"
                       % (best["batch"], best["threads"], best["placement"], best["steps"], best["sec_per_step"],
                          best["sec_per_step_std"], best["warmup"], head["threads"], head["placement"]),
                runs=runs, host=dict(logical_cpus=os.cpu_count(), physical_cores=phys, model=cpu_model(),
                                     socket0_cores=len(socket0) if socket0 else None))


def gather_bandwidth():
    """north_star: HBM GB/s on the embedding gather.  The step's own gather moves 19.7 MB (too small to show
    bandwidth), so it is timed at B*T = 2^20 tokens: D = 300 fp32 rows out of the 10 001-row table
    written time-major.  What HBM sees: the 1.26 GB of row stores and the 8 MB of ids -- the 12 MB table is
    cache resident, so its reads are NOT HBM traffic; `hbm_*` count only the former, `alg_*` is SURVEY 8d's
    algorithmic byte count (row read + row write + id) kept for reference."""
    import torch
    from tumblr_emotions_amd import ops
    V, D, B, T = 10000, 300, 8192, 128
    table = torch.randn(V + 1, D, device="cuda")
    ids = torch.randint(0, V + 1, (B, T), device="cuda", dtype=torch.int64)
    out = torch.empty(T * B, D, device="cuda")
    for _ in range(3):
        ops.gather_rows(table, ids, out, B, T, D)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gather_rows(table, ids, out, B, T, D)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    hbm = B * T * (D * 4 + 8)                           # row stores + ids
    alg = B * T * (2 * D * 4 + 8)                       # read row + write row + id
    return dict(kernel="gather_rows_kernel", tokens=B * T, us=round(us, 1), hbm_bytes=hbm,
                hbm_GBps=round(hbm / us / 1e3, 1), peak_GBps=8000, frac=round(hbm / us / 1e3 / 8000, 4),
                alg_bytes=alg, alg_GBps=round(alg / us / 1e3, 1),
                note="frac = HBM-visible bytes (row stores + ids) / time / 8 TB/s; the table reads are served by the caches")


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def conv_arithmetic(net):
    """Which multiply each conv launch of the image tower runs in, as ds_conv_plan decided (forward / dgrad launches per
    kernel family).  For --dtype fp8 this is the honest label: fp8 only on the layers where ds_conv_fp8 beats the bf16
    kernels (profiles/r04_fp8_layers_b128.txt), bf16 elsewhere."""
    img = getattr(net, "image", None)
    if img is None:
        return None
    from tumblr_emotions_amd import ops
    names = {getattr(ops, k): k[7:].lower() for k in dir(ops) if k.startswith("DS_FAM_")}
    count = {"forward": {}, "dgrad": {}}
    for layer in img.layers:
        for role, plan in (("forward", getattr(layer, "fwd", None)), ("dgrad", getattr(layer, "dgrad", None))):
            fam = getattr(plan, "family", None)
            if fam is not None:
                key = names.get(fam, str(fam))
                count[role][key] = count[role].get(key, 0) + 1
    label = {"f32": "fp32 MFMA", "bf16": "bf16 MFMA (3x3 input gradients: F(4x4) of the bf16-rounded operands, two bf16 pieces per Winograd-domain value)",
             "fp8": "bf16 MFMA + fp8 MFMA where it wins (bf16+fp8-auto)"}[img.dtype]
    return dict(label=label, launches_by_family=count)


def self_launch_argv(gpus, argv, port=None):
    """The command `python bench.py --gpus N ...` turns itself into when it was started WITHOUT a launcher (no WORLD_SIZE in
    the environment): one rank per GPU under torch.distributed.run on this node, rendezvous on 127.0.0.1 (the container
    hostname may not resolve), the caller's own arguments passed through unchanged.  Data-parallel semantics are those of
    /root/reference/slim/deployment/model_deploy.py:221-223,301-302; rank 0 still prints the ONE JSON line."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


CONV_FAMILY = ("conv_igemm_kernel", "conv_glds_kernel", "gemm_wide_kernel", "conv_wino_kernel", "conv_wino4_kernel",
               "conv_stem_kernel", "conv_bf16_kernel", "conv_bf16d_kernel", "conv_fp8d_kernel")


def pmc_traffic_live(steps=3, timeout=90):
    """HBM bytes per conv-family launch measured NOW: two rocprofv3 sub-runs of this same bench (default workload,
    `steps` steps, Mixed-block branches on one stream as in the roofline timing pass), one per counter -- FETCH_SIZE and
    WRITE_SIZE in SEPARATE passes with --kernel-trace only, FETCH doubled for gfx950, exactly as MI355X_MICROARCH.md
    prescribes -- summed over the launches the roofline brackets.  Returns (bytes per launch, description) or None when
    rocprofv3 is missing, a pass fails or times out (the caller then falls back to the committed profile)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    tot, calls = {}, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="ds_pmc_", dir="/tmp")
            env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-conv-timing",
                   "--no-gather", "--no-branch-streams", "--no-live-traffic"]
            # own process group: on a timeout the profiler AND the bench it wraps are killed, nothing lingers on the GPU
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                raise
            if rc != 0:
                raise RuntimeError("rocprofv3 exited with %d" % rc)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            cur = sqlite3.connect(dbs[0]).cursor()
            n, b = 0, 0.0
            for name, val in cur.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
                short = name.replace("(anonymous namespace)::", "").replace("void ", "", 1)
                if short.startswith(CONV_FAMILY):
                    n += 1
                    b += val * 1024.0          # KB -> bytes
            shutil.rmtree(d, ignore_errors=True)
            if n == 0:
                return None
            tot[counter], calls[counter] = b, n
        if calls["FETCH_SIZE"] != calls["WRITE_SIZE"]:
            raise RuntimeError("the two counter passes saw different launch counts: %r" % (calls,))
        calls = calls["FETCH_SIZE"]
        hbm = 2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]
        return round(hbm / calls), ("live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only; FETCH x 2 "
                                    "for gfx950) over 1 warm-up + %d steps of this workload, %d bracketed launches" % (steps, calls))
    except Exception as e:      # missing counters, a crashed pass, a timeout: never fail the benchmark over this
        sys.stderr.write("bench.py: live PMC traffic pass failed (%s); using the committed profile\n" % (e,))
        return None


def pmc_traffic(args):
    """HBM bytes per conv-family launch: measured live (pmc_traffic_live) for the default workload, else read from the
    committed rocprofv3 PMC passes of THIS workload (profiles/rNN_pmc.json, made by scripts/pmc_summary.py the same
    way), else null.  `traffic_source` says which."""
    import glob
    if args.batch != 256 or args.mode != "joint" or args.gpus != 1 or args.train_all or args.dtype != "f32" or args.mul3:
        return None, None
    if not args.no_live_traffic:
        live = pmc_traffic_live()
        if live is not None:
            return live
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    return round(d["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: fixed global batch split over the ranks (BASELINE cfg4: 256 over 8 GPUs)")
    ap.add_argument("--mode", default="joint", choices=["joint", "image", "text"])
    ap.add_argument("--train-all", action="store_true",
                    help="optional full fine-tuning (not the BASELINE workload): every conv weight and the "
                         "embedding trainable; 9.032 GFLOP/sample joint (SURVEY 8d)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "fp8"],
                    help="f32: the headline (exact fp32 MFMA, 1e-3 parity path).  bf16: SEPARATE, labelled line -- conv "
                         "forward/dgrad multiplies on the bf16 matrix pipe, fp32 storage/accumulate/statistics/masters")
    ap.add_argument("--stepwise-lstm", action="store_true",
                    help="A/B aid: one GEMM + one cell launch per LSTM step instead of the persistent ds_lstm_seq kernels")
    ap.add_argument("--lstm-rows", type=int, default=0,
                    help="row groups per workgroup of the persistent LSTM kernels (1, 2, 4, 8; default: the net's choice)")
    ap.add_argument("--side-mode", type=int, default=-1,
                    help="A/B: 0 = Branch_2 and Branch_3 chains on a side stream each, 1 = both on one, 2 = only Branch_3 (default: 0 up to 32 samples per GPU, else 1)")
    ap.add_argument("--no-pool-first", action="store_true",
                    help="Mixed backward: fused 1x1 dgrad writes the block-input gradient and the Branch_3 pool adds (default: the reverse)")
    ap.add_argument("--no-stem-direct", action="store_true",
                    help="Conv2d_1a_7x7 through the generic kernel on a 4-channel copy (default: ds_conv_stem on the packed RGB batch)")
    ap.add_argument("--no-bwd-sums", action="store_true",
                    help="A/B aid: BatchNorm backward sums by the separate ds_bn_bwd_reduce pass everywhere (default: from the "
                         "epilogue of the producing dgrad where the kernel can, DS_EPI_BNSUMS)")
    ap.add_argument("--no-branch-streams", action="store_true",
                    help="Mixed blocks on one stream (default: Branch_2 / Branch_3 on side streams)")
    ap.add_argument("--bf16-staged", action="store_true",
                    help="--dtype bf16: the LDS-staged bf16 kernel for every conv (default: ds_conv_bf16 where it wins)")
    ap.add_argument("--no-act16", action="store_true",
                    help="--dtype bf16 / fp8: keep the activations in fp32 storage (default there: 16-bit activation storage)")
    ap.add_argument("--no-winograd", action="store_true",
                    help="A/B aid: implicit GEMM for every 3x3 layer instead of the fused Winograd F(2x2,3x3) kernel")
    ap.add_argument("--mul3", action="store_true",
                    help="separately labelled line (dtype f32x3): the forward 1x1 convs through ds_conv_f32x3 -- fp32 products "
                         "from three bf16 pieces per operand on the bf16 matrix cores (fp32-MFMA accuracy, not its bits)")
    ap.add_argument("--no-zcat", action="store_true",
                    help="A/B aid: every conv followed by its BatchNorm-apply pass (default: the 3x3 / Branch_3 convs of "
                         "Mixed_3b..4f write z into the concat and the consumers normalise on load)")
    ap.add_argument("--no-fuse-b3", action="store_true",
                    help="A/B aid: Branch_3's 3x3/1 max pool as its own pass in front of the 1x1 conv (default: formed on load "
                         "by the conv, ds_conv_desc.pool_argmax)")
    ap.add_argument("--no-stem-pool", action="store_true",
                    help="A/B aid: Conv2d_1a_7x7 writes its full-resolution output and MaxPool_2a runs as its own pass (default: "
                         "the pool inside the stem kernel, ds_conv_stem_pool)")
    ap.add_argument("--no-wino4", action="store_true",
                    help="A/B aid: F(2x2,3x3) also on the 56 x 56 / 28 x 28 maps instead of the F(4x4,3x3) kernel")
    ap.add_argument("--graph", action="store_true",
                    help="replay the training step as one captured hipGraph (single rank; what bounds small per-GPU batches "
                         "is host launch cost)")
    ap.add_argument("--serial-towers", action="store_true",
                    help="A/B aid: text tower on the main stream instead of concurrently with the image tower")
    ap.add_argument("--rccl-world1", action="store_true",
                    help="--gpus 1 only: initialise the nccl (= RCCL) backend with ONE rank and take the bucketed all-reduce path "
                         "(early bucket 1 on the side stream, async work handle) exactly as a multi-GPU run does; prints the `dp` "
                         "block.  What a 1-GPU box can show of the 8-GPU code path")
    ap.add_argument("--no-overlap-comm", action="store_true",
                    help="data parallel A/B: both gradient buckets all-reduced after the backward pass on the main stream "
                         "(default: bucket 1 launched early on a side stream, under the Inception backward)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--no-conv-timing", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the embedding-gather bandwidth measurement")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed profile instead of two rocprofv3 --pmc sub-runs (about 40 s)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (how the driver starts the N = 1 bench): become the N-rank job
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # ROCr reads it at hsa_init: before any rank starts
        cmd = self_launch_argv(args.gpus, sys.argv[1:])
        sys.stderr.write("bench.py: no launcher in the environment, re-executing as: %s\n" % " ".join(cmd))
        sys.stderr.flush()
        os.execv(cmd[0], cmd)

    import torch
    import torch.distributed as dist
    from tumblr_emotions_amd import ops
    from tumblr_emotions_amd.net import SentimentNet
    from tumblr_emotions_amd.synthetic import synthetic_batch_numpy, to_device

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    # DS_BENCH_ONE_DEVICE=1 (self-test on a 1-GPU box): every rank uses cuda:0 and gloo carries the
    # all-reduce, because RCCL refuses two ranks on one device.  Never set for a real measurement.
    one_device = os.environ.get("DS_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    from tumblr_emotions_amd import dp, streams
    force_dp = bool(args.rccl_world1 and world == 1)
    if world > 1 or force_dp:
        # dp.init_distributed: the step's streams take their hardware queues BEFORE RCCL creates its own
        if force_dp:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dp.init_distributed("nccl", device=dev_index, rank=0, world_size=1)
        else:
            dp.init_distributed("gloo" if one_device else "nccl", device=dev_index)
    elif os.environ.get("DS_BENCH_NO_RESERVE") != "1":
        streams.reserve()
    rccl_ranks = None
    if world > 1 or force_dp:          # a real collective before anything is timed: how many ranks does the backend connect?
        probe = torch.ones(1, device="cuda")
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        rccl_ranks = int(round(float(probe.item())))

    T, V, D, H = 32, 10000, 300, 512
    net = SentimentNet(mode=args.mode, nb_emotions=15, im_features_size=256, rnn_size=H, fc_size=512, vocab_size=V,
                       embedding_dim=D, post_size=T, dropout_keep_prob=0.8, train_all=args.train_all,
                       trainable_embedding=args.train_all, concurrent_towers=not args.serial_towers,
                       dtype=args.dtype, force_dp_buckets=force_dp and os.environ.get("DS_BENCH_INIT_ONLY") != "1",
                       overlap_comm=not args.no_overlap_comm)
    net.initialize(seed=1)
    if args.lstm_rows and net.text is not None:
        net.text.seq_rows = args.lstm_rows
    if args.stepwise_lstm and net.text is not None:
        net.text.persistent = False
    if args.no_winograd and net.image is not None:
        net.image.winograd = False
    if args.no_wino4 and net.image is not None:
        net.image.winograd4 = False
    if args.no_zcat and net.image is not None:
        net.image.zcat = False
    if args.no_fuse_b3 and net.image is not None:
        net.image.fuse_branch3 = False
    if args.no_stem_pool and net.image is not None:
        net.image.stem_pool = False
    if args.mul3 and net.image is not None:
        net.image.mul3 = True
    if args.no_branch_streams and net.image is not None:
        net.image.branch_streams = False
    if args.side_mode >= 0 and net.image is not None:
        net.image.side_mode = args.side_mode
    if os.environ.get("DS_LIB") and os.environ.get("DS_WGRAD_SIDE") == "0" and net.image is not None:
        net.image.wgrad_side = False
    if args.no_bwd_sums and net.image is not None:
        net.image.bwd_sums = False
    if args.no_pool_first and net.image is not None:
        net.image.pool_first = False
    if args.no_stem_direct and net.image is not None:
        net.image.stem_direct = False
    if args.bf16_staged and net.image is not None:
        net.image.bf16_direct = False
        net.image.act16 = False
    if args.no_act16 and net.image is not None:
        net.image.act16 = False
    strong = args.global_batch > 0
    if strong:
        assert args.global_batch % world == 0, "--global-batch must divide by the number of ranks"
        args.batch = args.global_batch // world
    gb = args.batch * world
    batch = to_device(synthetic_batch_numpy(gb, T, V, 15, seed=0, with_images=args.mode != "text"), "cuda", rank, world)
    lr = 1e-3

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    graphed = bool(args.graph and world == 1 and net.capture_step(batch))
    for _ in range(args.warmup):
        net.train_step(batch, lr)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        net.train_step(batch, lr)
    t_enq = time.perf_counter() - t0          # host time to ENQUEUE the K steps (the GPU runs behind it)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss = net.total_loss_value()
    dp_report = None
    if world > 1 or force_dp:          # two more steps with event pairs around the early bucket-1 reduce (not in the timed region)
        net.reducer.timing = True
        for _ in range(2):
            net.train_step(batch, lr)
        barrier()
        dp_report = net.reducer.overlap_report() or {}
        dp_report.update(rccl_ranks=rccl_ranks, backend=dist.get_backend(), world_size=dist.get_world_size(),
                         overlap_enabled=bool(net.reducer.overlap))
        net.reducer.timing = False

    # Roofline pass: the SAME K steps again, now with a HIP event pair around every conv/GEMM launch on
    # the stream it is launched on.  Kept out of the headline region because the 2 x 148 event records per
    # step cost 2-3 % of the step (measured); everything else is identical (towers concurrent, all-reduce).
    timer = None
    dt_events = None
    if graphed:
        net.release_graph()      # the roofline passes time individual launches: eager
    in_situ = None
    if not args.no_conv_timing:
        # Event-bracketed launch times are a kernel's own duration only when nothing else is queued beside it: the
        # timing pass therefore issues the Mixed-block branches on ONE stream (the headline pass runs Branch_2 /
        # Branch_3 on side streams, where a launch's bracket also contains the time it waited for CUs).  The
        # concurrent figure is reported next to it as achieved_branches_concurrent.
        img = net.image
        concurrent = img is not None and img.branch_streams
        if concurrent and world == 1:
            t0 = ops.ConvTimer()
            ops.CONV_TIMER = t0
            for _ in range(3):
                net.train_step(batch, lr)
            torch.cuda.synchronize()
            ops.CONV_TIMER = None
            n0, ms0, fl0 = t0.summary()
            in_situ = round(fl0 / (ms0 * 1e-3) / 1e12, 2)
        if concurrent:
            img.branch_streams = False
        timer = ops.ConvTimer()
        ops.CONV_TIMER = timer
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            net.train_step(batch, lr)
        barrier()
        dt_events = time.perf_counter() - t1
        ops.CONV_TIMER = None

    # In a step the text tower's kernels run concurrently with the Inception kernels on a second stream,
    # so a conv launch's event-timed duration includes the time it shared the CUs.  For reference also
    # time the same launches with the towers serialised (3 more steps).
    isolated = None
    if timer is not None and getattr(net, "text_stream", None) is not None and world == 1:
        side, net.text_stream = net.text_stream, None
        t2 = ops.ConvTimer()
        ops.CONV_TIMER = t2
        for _ in range(3):
            net.train_step(batch, lr)
        torch.cuda.synchronize()
        ops.CONV_TIMER = None
        net.text_stream = side
        n2, ms2, fl2 = t2.summary()
        isolated = round(fl2 / (ms2 * 1e-3) / 1e12, 2)
    if not args.no_conv_timing and net.image is not None and not args.no_branch_streams:
        net.image.branch_streams = True

    # BASELINE configs[3] names a GLOBAL batch of 256 split over the GPUs (strong scaling); the headline above keeps 256 per
    # GPU (weak scaling, the contract's `value`).  A multi-rank run therefore times BOTH in one invocation: the same model
    # re-allocated at 256 / world samples per rank, W warm-up + K timed steps bracketed like the headline, its own `dp` block.
    strong_block = None
    if world > 1 and not strong and 256 % world == 0 and os.environ.get("DS_BENCH_NO_STRONG") != "1":
        sb_n = 256 // world
        sbatch = to_device(synthetic_batch_numpy(256, T, V, 15, seed=0, with_images=args.mode != "text"), "cuda", rank, world)
        for _ in range(args.warmup):
            net.train_step(sbatch, lr)
        barrier()
        ts = time.perf_counter()
        for _ in range(args.steps):
            net.train_step(sbatch, lr)
        barrier()
        dts = time.perf_counter() - ts
        tmax = torch.tensor([dts], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dts = float(tmax.item())
        net.reducer.timing = True
        for _ in range(2):
            net.train_step(sbatch, lr)
        barrier()
        sdp = net.reducer.overlap_report() or {}
        net.reducer.timing = False
        strong_block = dict(metric="training samples/sec (224x224 img + 32-tok text, GLOBAL batch 256 = BASELINE configs[3])",
                            value=round(256 * args.steps / dts, 2), unit="samples/s", scaling="strong", n_gpus=world,
                            global_batch=256, per_gpu_batch=sb_n, steps=args.steps, warmup=args.warmup,
                            ms_per_step=round(1e3 * dts / args.steps, 3), dp=sdp)

    if rank == 0:
        value = gb * args.steps / dt
        flop_per_sample = {"joint": GFLOP_PER_SAMPLE, "image": 5.885, "text": 0.280}[args.mode]
        if args.train_all:            # whole tower unfrozen; the embedding's dX GEMM adds 0.039
            flop_per_sample = {"joint": 9.032 + 0.039, "image": 8.748, "text": 0.280 + 0.039}[args.mode]
        roof = None
        if timer is not None:
            n, ms, flops = timer.summary()
            ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            n_p, ms_p, fl_p = timer.summary(pooled=True)           # launches that also carry a max pool (round 6)
            n_c, ms_c, fl_c = timer.summary(pooled=False)
            traffic, traffic_src = pmc_traffic(args)
            peak = {"bf16": PEAK_BF16_MFMA_TFLOPS, "fp8": PEAK_FP8_MFMA_TFLOPS}.get(args.dtype, PEAK_FP32_MFMA_TFLOPS)
            kname = ("conv_bf16_kernel (ds_conv_igemm, DS_DTYPE_BF16: v_mfma_f32_32x32x16_bf16 implicit GEMM, fp32 accumulate; conv "
                     "fwd + dgrad) together with the fp32 GEMMs of the LSTM / heads launched through the same entry point"
                     if args.dtype == "bf16" else
                     "conv_fp8d_kernel (ds_conv_fp8: v_mfma_f32_32x32x16_fp8_fp8 forward, _bf8_fp8 dgrad, per-tensor power-of-two "
                     "scales, fp32 accumulate) for the 1x1 / 3x3 convs, bf16 stem, together with the fp32 GEMMs of the LSTM / heads"
                     if args.dtype == "fp8" else
                     "conv_igemm_kernel + conv_glds_kernel + gemm_wide_kernel + conv_stem_kernel + conv_wino_kernel + conv_wino4_kernel "
                     "(fp32 v_mfma_f32_32x32x2_f32: implicit GEMM for conv fwd / dgrad / GEMMs through ds_conv_igemm and ds_conv_stem, "
                     "fused Winograd F(4x4,3x3) / F(2x2,3x3) for the 3x3 layers through ds_conv_wino4 / ds_conv_wino; FLOPs counted "
                     "are the convolution's 2*M*N*K, so the Winograd launches can exceed the matrix peak)")
            roof = dict(bound="mfma", kernel=kname,
                        achieved=round(ach, 2), peak=peak, unit="TFLOP/s",
                        frac=round(ach / peak, 4), traffic=traffic, traffic_unit="HBM bytes per launch",
                        traffic_source=traffic_src, alg_flops_per_launch=round(flops / max(n, 1)),
                        achieved_towers_serialised=isolated, achieved_branches_concurrent=in_situ,
                        launches_per_step=n // max(args.steps, 1), avg_launch_us=round(1e3 * ms / max(n, 1), 2),
                        kernel_time_share=round(ms * 1e-3 / dt_events, 3),
                        timing_pass="second pass of the same %d steps with HIP events around every launch and the "
                                    "Mixed-block branches on one stream, so a bracket is the launch's own duration "
                                    "(%.3f ms/step in that pass); the headline pass carries no events and runs the "
                                    "branches on a side stream" % (args.steps, 1e3 * dt_events / args.steps),
                        whole_step_tflops=round(value * flop_per_sample / 1e3, 2),
                        whole_step_frac=round(value * flop_per_sample / 1e3 / peak, 4),
                        launches_with_a_pool_inside=dict(
                            per_step=n_p // max(args.steps, 1), avg_launch_us=round(1e3 * ms_p / max(n_p, 1), 2),
                            achieved=round(fl_p / (ms_p * 1e-3) / 1e12, 2) if ms_p > 0 else None,
                            achieved_of_the_other_launches=round(fl_c / (ms_c * 1e-3) / 1e12, 2) if ms_c > 0 else None,
                            note="since round 6 the stem launch contains MaxPool_2a and the Branch_3 1x1 launches of Mixed_3b..5b "
                                 "contain their 3x3/1 max pool (pool passes that used to be separate, non-MFMA launches): `achieved` "
                                 "and `frac` above divide the same convolution FLOPs by conv + pool time; the other launches alone "
                                 "run at `achieved_of_the_other_launches`"))
        out = {
            "metric": "training samples/sec (224x224 img + 32-tok text, batch 256)",
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "sec_per_step": round(dt / args.steps, 5), "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": ("f32x3" if args.mul3 and args.dtype == "f32" else args.dtype), "data": "synthetic",
            "config": {"workload": "%s train step (fwd+bwd+all-reduce+Adam): Inception-v1 224x224x3 + 300-d embedding + "
                                   "LSTM-512, T=32, V=10000, 15 classes, batch %d per GPU, %s, dropout 0.8, BN train mode"
                                   % (args.mode, args.batch,
                                      "FULL FINE-TUNING (every conv weight and the embedding trainable: NOT the "
                                      "BASELINE workload)" if args.train_all else
                                      "reference freeze (<=Mixed_5b conv weights frozen, all BN betas trainable)")
                                   + (", conv fwd/dgrad multiplies in bf16 (fp32 storage, accumulation, statistics, master "
                                      "weights; NOT the fp32 parity configuration)" if args.dtype == "bf16" else
                                      ", 1x1 / 3x3 conv fwd (e4m3 x e4m3) and dgrad (e5m2 x e4m3) multiplies on the fp8 matrix pipe with "
                                      "per-tensor power-of-two scales, bf16 stem (fp32 storage, accumulation, statistics, master weights; "
                                      "BASELINE configs[4]'s conv path, NOT the fp32 parity configuration)" if args.dtype == "fp8" else ""),
                       "global_batch": gb, "per_gpu_batch": args.batch, "parallelism": "dp%d" % world,
                       "gflop_per_sample": flop_per_sample, "final_loss": round(loss, 5),
                       "launch": "hipGraph replay" if graphed else "eager",
                       "host_enqueue_ms_per_step": round(1e3 * t_enq / args.steps, 3)},
            "roofline": roof,
            "conv_arithmetic": conv_arithmetic(net),
            "dp": dp_report,
            "strong_scaling": strong_block,
            "gather": gather_bandwidth() if (args.mode != "image" and not args.no_gather) else None,
        }
        if world == 1 and not args.no_cpu_baseline and args.mode == "joint" and not args.train_all:
            out["cpu_baseline"] = cpu_baseline(T, V, D, H, args.cpu_warmup, args.cpu_steps)
        else:
            out["cpu_baseline"] = None
    if world > 1:
        barrier()            # rank 0 is still timing the gather / the CPU baseline: leave together
    if world > 1 or force_dp:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio, which sits in a
        # buffer until it is flushed -- flush it first
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
