#!/bin/bash
# conv_wino4: the launch-time model's NB choice against forced NB = 1 / 2 on the 14x14 and 7x7 layers (B = 256, 32)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06u
S="14x96x208 14x112x224 14x128x256 14x144x288 14x160x320 14x16x48 14x32x64 14x32x128 7x160x320 7x192x384 7x32x128 7x48x128 28x96x128 28x128x192 28x16x32 28x32x96"
for B in 256 32; do
for nb in 0 1 2; do echo "== B=$B DS_WINO4_NB=$nb"; DS_WINO4_NB=$nb python scripts/wino4_bench.py $B $S 2>&1 | grep -v "amdgpu.ids\|overrides"; done
done > gpurun_out/r06u/nb.txt 2>&1
python - <<'PY'
import re, collections
cur = None; d = collections.defaultdict(dict)
for l in open("gpurun_out/r06u/nb.txt"):
    m = re.match(r"== B=(\d+) DS_WINO4_NB=(\d)", l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    a = l.split("|")
    if len(a) >= 5 and a[0].split()[0].isdigit():
        hw, ci, co = a[0].split()
        f4 = float(a[3].split()[0]); f2 = float(a[2].split()[0])
        key = (cur[0], hw, ci, co, "dgrad" if "dgrad" in l else "fwd")
        d[key][cur[1]] = f4; d[key]["f2"] = f2
print("%4s %3s %4s %4s %5s | %8s %8s %8s %8s" % ("B", "HW", "Cin", "Cout", "", "auto", "NB=1", "NB=2", "F(2x2)"))
tot = collections.defaultdict(lambda: [0, 0, 0, 0])
for k in sorted(d, key=lambda k: (-k[0], -int(k[1]), int(k[2]), k[4])):
    v = d[k]
    print("%4d %3s %4s %4s %5s | %8.1f %8.1f %8.1f %8.1f" % (k[0], k[1], k[2], k[3], k[4], v.get(0, 0), v.get(1, 0), v.get(2, 0), v["f2"]))
    t = tot[k[0]]; t[0] += v.get(0, 0); t[1] += v.get(1, 0); t[2] += v.get(2, 0); t[3] += min(v.get(1, 1e9), v.get(2, 1e9))
for b, t in tot.items(): print("B=%d sum: auto %.1f  NB=1 %.1f  NB=2 %.1f  best-of-two %.1f" % (b, *t))
PY
