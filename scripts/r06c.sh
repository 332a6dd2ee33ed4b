#!/bin/bash
mkdir -p gpurun_out/r06c
python scripts/stem_bench.py 256 > gpurun_out/r06c/stem_bench.txt 2>&1
python scripts/stem_bench.py 32 >> gpurun_out/r06c/stem_bench.txt 2>&1
KT_LINES=70 bash scripts/ktrace.sh r06c/serial --no-branch-streams > /dev/null 2>&1
KT_LINES=70 bash scripts/ktrace.sh r06c/serial_old --no-branch-streams --no-stem-pool --no-fuse-b3 > /dev/null 2>&1
cat gpurun_out/r06c/stem_bench.txt
