#!/bin/bash
mkdir -p gpurun_out/r06i
for r in 0 3 4; do DS_WIDE_RING=$r python scripts/wide_epi_bench.py 256 > gpurun_out/r06i/ring$r.txt 2>&1; done
paste <(cut -c1-60 gpurun_out/r06i/ring0.txt) <(cut -c24-60 gpurun_out/r06i/ring3.txt) <(cut -c24-60 gpurun_out/r06i/ring4.txt) | head -40
for r in 0 3 4; do grep -c . gpurun_out/r06i/ring$r.txt; done
diff <(awk '{print $NF}' gpurun_out/r06i/ring0.txt) <(awk '{print $NF}' gpurun_out/r06i/ring3.txt) | head -5; echo "fp diff 0 vs 3 done"
diff <(awk '{print $NF}' gpurun_out/r06i/ring0.txt) <(awk '{print $NF}' gpurun_out/r06i/ring4.txt) | head -5; echo "fp diff 0 vs 4 done"
