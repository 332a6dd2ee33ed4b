"""The HIP streams of the training step, created and first used in a FIXED order.

ROCm gives a process a small number of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and binds a hipStream to one of
them round-robin when the stream first submits work.  Two streams on one hardware queue serialise.  The step keeps four
streams busy at once (main, the Mixed-block side stream, the text tower's stream, the gradient all-reduce stream);
measured on MI355X: with the RCCL communicator initialised BEFORE these streams first ran, its internal streams had
taken queue slots and the text tower's stream landed on the main stream's queue -- the towers serialised and the step
went 17.7 -> 19.1 ms although the all-reduce itself took 0.03 ms (profiles/r03_notes.md).  Reserving the step's streams
-- create, submit one tiny kernel each, in this order -- before anything else creates streams keeps them on distinct
queues whether or not a process group exists.  A launcher that uses torch.distributed must call `reserve()` right
after `torch.cuda.set_device(...)` and BEFORE `torch.distributed.init_process_group(...)`: `dp.init_distributed()` does
exactly that and is what bench.py and the multi-process tests use.  The train_* entry points do not create process
groups; `SentimentNet.__init__` calls `reserve()` as well, which is early enough only while no process group exists.
The table is per device: a second engine on another device gets its own streams.
"""
import os

import torch

from . import _lib

_ORDER = ("side1", "text", "comm", "side0")
_by_device = {}      # device index -> {role: stream}
# (A/B aid, measured neutral: DS_TEXT_PRIO=1 gives the text stream high priority)
_PRIORITY = {"text": -1 if _lib.tuning_env("DS_TEXT_PRIO", "0") == "1" else 0}


def _index(device):
    if device is None:
        return torch.cuda.current_device()
    dev = torch.device(device)
    return torch.cuda.current_device() if dev.index is None else dev.index


def reserve(device=None):
    """Create the step's streams on `device` (default: the current one) and bind each to a hardware queue now
    (idempotent per device)."""
    if not torch.cuda.is_available():
        return
    idx = _index(device)
    if idx in _by_device:
        return
    _streams = _by_device[idx] = {}
    dev = torch.device("cuda", idx)
    touch = torch.zeros(64, device=dev)
    touch.add_(1.0)                                   # the main (current) stream submits first
    plan = _lib.tuning_env("DS_STREAM_PLAN")           # experiment aid: creation order, "x" = a dummy stream
    order = tuple(plan.split(",")) if plan else _ORDER
    for name in order:
        s = torch.cuda.Stream(device=dev, priority=_PRIORITY.get(name, 0))
        with torch.cuda.stream(s):
            touch.add_(1.0)                           # first submission: the stream takes its hardware queue
        if name == "x":
            _streams.setdefault("_dummies", []).append(s)
        else:
            _streams[name] = s
    for name in _ORDER:
        if name not in _streams:
            _streams[name] = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)


def get(name, device=None):
    """The reserved stream of `device` (default: the current one) for a role: 'side0' / 'side1' (Mixed-block branch
    chains), 'text' (text tower), 'comm' (gradient all-reduce)."""
    idx = _index(device)
    if idx not in _by_device:
        reserve(idx)
    return _by_device[idx][name]
