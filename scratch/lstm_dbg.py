import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tumblr_emotions_amd import ops
def run(B, T, H, rows, seed=0):
    rng = np.random.RandomState(B + T + H)
    pre = torch.tensor(rng.normal(size=(T, B, 4 * H)) * 0.7, dtype=torch.float32).cuda()
    wh = torch.tensor(rng.normal(size=(H, 4 * H)) * (0.5 / np.sqrt(H)), dtype=torch.float32).cuda()
    seq = rng.randint(1, T + 1, size=B).astype(np.int64); seq[0], seq[-1] = 1, T
    dh = torch.tensor(rng.normal(size=(B, H)), dtype=torch.float32).cuda()
    seqd = torch.from_numpy(seq).cuda()
    ws = torch.zeros(max(ops.lstm_seq_workspace(B, H) // 4, 4), dtype=torch.int32, device="cuda")
    h = torch.zeros(T + 1, B, H, device="cuda"); c = torch.zeros(T + 1, B, H, device="cuda")
    ops.lstm_seq_set_rows(rows)
    g = pre.clone()
    ops.lstm_seq_fwd(g, ops._p(wh), 4 * H, h, c, seqd, T, B, H, 1.0, ws)
    dg = torch.full((T, B, 4 * H), float("nan"), device="cuda")
    ops.lstm_seq_bwd(g, ops._p(wh), 4 * H, c, dh, dh.stride(0), seqd, T, B, H, dg, ws)
    torch.cuda.synchronize()
    ops.lstm_seq_set_rows(1)
    return h, c, g, dg
for (B, T, H) in [(37, 9, 64), (64, 12, 128), (70, 9, 64), (37, 9, 32), (256, 32, 512)]:
    ref = run(B, T, H, 1)
    for rows in (2, 2, 4, 8):
        out = run(B, T, H, rows)
        d = (out[3] - ref[3]).abs()
        bad = torch.nonzero(d > 0)
        print(B, T, H, "rows", rows, "fwd equal", all(torch.equal(a, b) for a, b in zip(out[:3], ref[:3])), "bwd maxdiff %.3e" % float(d.max()), "nbad", len(bad),
              "t/b/col of first", bad[0].tolist() if len(bad) else None, "b range", (int(bad[:, 1].min()), int(bad[:, 1].max())) if len(bad) else None,
              "t range", (int(bad[:, 0].min()), int(bad[:, 0].max())) if len(bad) else None)
