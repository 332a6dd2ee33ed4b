"""Data parallelism for the Deep Sentiment step: one process per GPU, RCCL over xGMI.

The reference has no distributed code of its own; the only data-parallel semantics in the tree are
slim's in-graph clones (slim/deployment/model_deploy.py): per-clone loss divided by num_clones
(:221-223), clone gradients summed (:414-444), the regularisation term counted once (:301-302),
BatchNorm statistics and moving averages per clone (:353-355).  Here that becomes:

  * rank r trains on the r-th contiguous slice of the same seeded global batch;
  * gradients of all trainable variables lie in ONE flat fp32 buffer (params.ParamStore), laid out
    [ bucket 1: L2-regularised conv weights | Logits bias, LSTM, heads, Mixed_5c betas ][ bucket 2:
    BatchNorm betas below Mixed_5c ];
  * bucket 1 (~14.7 MB at the BASELINE dims) is complete as soon as Mixed_5c's backward is done, so
    its sum-all-reduce is launched right there on a side stream and overlaps the remaining ~90 % of
    the Inception backward (dgrad through 5b ... 2b); bucket 2 (~25 KB) is reduced at the end;
  * the 1/world scale and the L2 gradient are applied inside the fused Adam kernel, after the
    reduction (L2 counted once, like slim);
  * BatchNorm uses per-rank batch statistics (slim keeps them per clone).

Nothing here needs a GPU: the same functions run under gloo on CPU tensors (tests/test_dp_cpu.py).
"""
import torch
import torch.distributed as dist


def init_distributed(backend="nccl", device=None, **kwargs):
    """`torch.distributed.init_process_group` for one rank of a data-parallel job, with the step's HIP streams bound to
    their hardware queues FIRST (streams.reserve: RCCL's own streams must not take the queue the text tower's stream
    would have got -- 1.4 ms per step, profiles/r03_notes.md).  `device` (index or torch.device; default: the current
    one) is made current here.  Extra keyword arguments go to init_process_group (rank, world_size, init_method ...).
    Used by bench.py and tests/test_dp_gpu.py.

    Two things this function cannot repair and therefore reports:
      * HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the environment before the process first touches HIP (ROCr reads it at
        hsa_init; this driver stack only supports dmabuf IPC).  It is set here for child processes and for a HIP runtime
        that has not started yet; if HIP is already up without it, a RuntimeWarning says so.
      * a process group that already exists was created BEFORE the streams were reserved: RuntimeError."""
    import os
    import warnings
    if dist.is_initialized():
        raise RuntimeError("init_distributed: a process group already exists; call this INSTEAD of "
                           "init_process_group so that streams.reserve() runs before RCCL creates its streams")
    if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
        if torch.cuda.is_available() and torch.cuda.is_initialized() and backend == "nccl":
            warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY=0 was not set before HIP initialised; RCCL across processes may "
                          "fail with hipIpcGetMemHandle: invalid argument", RuntimeWarning)
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if torch.cuda.is_available():
        from . import streams
        if device is not None:
            # (a torch.device("cuda") without an index means the current device)
            dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
            device = dev.index if dev.index is not None else torch.cuda.current_device()
            torch.cuda.set_device(device)
        streams.reserve(device)
        if backend == "nccl" and "device_id" not in kwargs:
            kwargs["device_id"] = torch.device("cuda", torch.cuda.current_device())
    dist.init_process_group(backend, **kwargs)
    return world_info()


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_bounds(n, rank, world):
    """Contiguous slice [lo, hi) of a global batch of n samples owned by `rank`."""
    return rank * n // world, (rank + 1) * n // world


class GradientReducer:
    """Sum-all-reduce of the flat gradient in two buckets, the first one optionally launched early on
    a side stream (overlap with the rest of the backward pass)."""

    def __init__(self, flat_grad, n_bucket1, group=None, overlap=True, force_buckets=False):
        self.g, self.n1, self.group = flat_grad, int(n_bucket1), group
        self.rank, self.world = world_info(group)
        # force_buckets: take the bucketed collective path even with ONE rank (an initialised process group is
        # required).  On a 1-GPU box this runs the real RCCL all-reduce, its async work handle and the side-stream
        # ordering exactly as a multi-GPU run does (sum over one rank = identity, scale 1/1), which gloo cannot show.
        if force_buckets and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("force_buckets needs an initialised torch.distributed process group")
        self.active = self.world > 1 or bool(force_buckets)
        self.overlap = overlap and self.active and flat_grad.is_cuda
        if self.overlap:
            from . import streams
            self.stream = streams.get("comm", flat_grad.device)
        else:
            self.stream = None
        self._pending = None
        self._ready = set()
        self._needed = set()
        self._events = []
        # timing=True (bench.py): event pairs around the early bucket-1 reduce on the side stream and at the end of
        # the backward pass on the main stream, so the first multi-GPU run says whether the overlap is real
        self.timing = False
        self._t = None

    # -- early launch of bucket 1 ----------------------------------------------------------------
    def expect(self, *names):
        """Names of the backward stages that must have finished before bucket 1 is complete."""
        self._needed = set(names)

    def begin_step(self):
        self._ready.clear()
        self._events = []
        self._pending = None

    def stage_done(self, name):
        """Called by the engines from inside backward(); launches bucket 1 when the last stage reports."""
        if not self.active:
            return
        self._ready.add(name)
        if self.overlap:       # stages may run on different streams (text tower): remember where each finished
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._events.append(ev)
        if self.overlap and self._pending is None and self._needed and self._needed <= self._ready:
            for ev in self._events:
                self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                if self.timing:
                    self._t = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    self._t[0].record(self.stream)          # bucket-1 reduce enters the side stream
                if self.timing:
                    import time
                    h0 = time.perf_counter()
                self._pending = dist.all_reduce(self.g[:self.n1], op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True)
                if self.timing:
                    self._host_ms = 1e3 * (time.perf_counter() - h0)      # does the launch block the host?

    # -- end of backward -------------------------------------------------------------------------
    def finish(self):
        """Reduce whatever has not been reduced yet and make the current stream wait for all of it.
        Returns the scale (1/world) the optimiser must apply to the summed gradient."""
        if not self.active:
            return 1.0
        if self._pending is not None:
            main = torch.cuda.current_stream()
            if self.timing and self._t is not None:
                self._t[2].record(main)                     # the backward pass has been enqueued up to here
            with torch.cuda.stream(self.stream):
                self._pending.wait()                        # side stream waits for RCCL's stream
                if self.timing and self._t is not None:
                    self._t[1].record(self.stream)          # bucket-1 reduce complete
            main.wait_stream(self.stream)
        else:
            dist.all_reduce(self.g[:self.n1], op=dist.ReduceOp.SUM, group=self.group)
        if self.g.numel() > self.n1:
            dist.all_reduce(self.g[self.n1:], op=dist.ReduceOp.SUM, group=self.group)
        self._pending = None
        return 1.0 / self.world

    def overlap_report(self):
        """After a synchronise, for the last step run with timing=True: how long bucket 1's all-reduce took on the
        side stream and how much of it was still outstanding when the backward pass ended (exposed = not hidden)."""
        if not self._t:
            return None
        t0, t1, t2 = self._t
        return dict(bucket1_bytes=int(self.n1) * 4, bucket2_bytes=int(self.g.numel() - self.n1) * 4,
                    bucket1_allreduce_ms=round(t0.elapsed_time(t1), 3),
                    backward_after_launch_ms=round(t0.elapsed_time(t2), 3),
                    bucket1_exposed_ms=round(max(0.0, t2.elapsed_time(t1)), 3),
                    bucket1_launch_host_ms=round(getattr(self, "_host_ms", 0.0), 3))
