#!/usr/bin/env python
"""Where does a step of the persistent LSTM forward kernel spend its time?  Workgroup (0,0) stamps s_memtime at
its phase boundaries (ds_debug_lstm_seq_set_profile); prints the per-phase mean over the steps in microseconds.
    python scripts/lstm_phase_prof.py [B] [H] [T]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tuning  # noqa: F401,E402  (the -DDS_TUNING library: ds_debug_* switches, DS_* knobs)
import numpy as np
import torch
from tumblr_emotions_amd import _lib, ops

B, H, T = (int(a) for a in (sys.argv[1:4] + ["256", "512", "32"][len(sys.argv) - 1:]))
lib = _lib.load()
gates = torch.randn(T, B, 4 * H, device="cuda") * 0.5
wh = torch.randn(H, 4 * H, device="cuda") * 0.02
h = torch.zeros(T + 1, B, H, device="cuda")
c = torch.zeros(T + 1, B, H, device="cuda")
seq = torch.full((B,), T, dtype=torch.int64, device="cuda")
ws = torch.zeros(max(ops.lstm_seq_workspace(B, H) // 4, 4), dtype=torch.int32, device="cuda")
prof = torch.zeros(T, 8, dtype=torch.int64, device="cuda")
g0 = gates.clone()
for it in range(3):
    gates.copy_(g0)
    lib.ds_debug_lstm_seq_set_profile(ops._p(prof) if it == 2 else None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.lstm_seq_fwd(gates, ops._p(wh), 4 * H, h, c, seq, T, B, H, 1.0, ws)
    e1.record()
    torch.cuda.synchronize()
    launch_us = 1e3 * e0.elapsed_time(e1)
    print("launch %d: %.1f us" % (it, launch_us))
lib.ds_debug_lstm_seq_set_profile(None)
p = prof.cpu().numpy().astype(np.float64)
tick = None                 # s_memtime counts shader cycles: calibrated against the event-timed launch below
names = ["wait for h[t]", "A loads + MFMA + LDS write", "barrier", "reduce + cell + stores issued", "drain (vmcnt 0)",
         "barrier + arrive"]
tick = launch_us / float(p[-1, 6] - p[0, 0])      # us per s_memtime tick (the stamps span the launch)
print("s_memtime: %.0f MHz" % (1.0 / tick))
d = np.diff(p[:, :7], axis=1)[1:] * tick
for n, v in zip(names, d.mean(axis=0)):
    print("%-34s %7.2f us" % (n, v))
print("%-34s %7.2f us" % ("step (stamp 0 -> 0)", np.diff(p[:, 0]).mean() * tick))

# ---- the backward launch (stamps of the steps T - 2 ... 0: step T - 1 has no hand-off) -----------------------------------
dg = torch.empty(T, B, 4 * H, device="cuda")
dh = torch.randn(B, H, device="cuda")
profb = torch.zeros(T, 8, dtype=torch.int64, device="cuda")
for it in range(3):
    lib.ds_debug_lstm_seq_set_profile_bwd(ops._p(profb) if it == 2 else None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.lstm_seq_bwd(gates, ops._p(wh), 4 * H, c, dh, H, seq, T, B, H, dg, ws)
    e1.record()
    torch.cuda.synchronize()
    print("backward launch %d: %.1f us" % (it, 1e3 * e0.elapsed_time(e1)))
lib.ds_debug_lstm_seq_set_profile_bwd(None)
pb = profb.cpu().numpy().astype(np.float64)[::-1][1:]          # in walk order, without step T - 1
namesb = ["wait for dgates[t+1]", "A loads + MFMA + LDS write", "barrier", "reduce + gate gradients + stores", "drain (vmcnt 0)",
          "barrier + arrive"]
db = np.diff(pb[:, :7], axis=1) * tick
for n, v in zip(namesb, db.mean(axis=0)):
    print("%-34s %7.2f us" % (n, v))
print("%-34s %7.2f us" % ("step (stamp 0 -> 0)", np.diff(pb[:, 0]).mean() * tick))
