#!/bin/bash
# NOTE: the LDS-transposed epilogue this A/B switched (DS_BF16D_TPOSE) was measured slower and removed again (profiles/r06_notes.md,
# profiles/r06_bf16d_tpose_ab.txt); kept as the record of the measurement.
# conv_bf16d: output through the LDS-transposed epilogue (16-byte stores) against the direct per-lane stores (DS_BF16D_TPOSE=0)
R=$(cd $(dirname $0)/.. && pwd)
export DS_LIB=$R/tumblr_emotions_amd/libds_kernels_tuning.so
mkdir -p gpurun_out/r06zg
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -x -q -k "bf16 or f32x3 or three_bf16 or centred or reading_bf16" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -x -q -k "bf16 or fp8 or 16_bit or f32x3 or mul3 or z_storage" 2>&1 | tail -4
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-conv-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do for e in 1 0; do echo "bf16 tpose=$e $(DS_BF16D_TPOSE=$e run --dtype bf16)"; echo "fp8 tpose=$e $(DS_BF16D_TPOSE=$e run --dtype fp8)"; echo "bf16_B128 tpose=$e $(DS_BF16D_TPOSE=$e run --dtype bf16 --batch 128)"; echo "mul3 tpose=$e $(DS_BF16D_TPOSE=$e run --mul3)"; done; done > gpurun_out/r06zg/ab.txt 2>&1
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r06zg/ab.txt"):
    a = l.split()
    if len(a) == 3: d[(a[0], a[1])].append(float(a[2]))
for k in sorted(d): print(k, " ".join("%.3f" % v for v in d[k]), "median %.3f" % statistics.median(d[k]))
PY
