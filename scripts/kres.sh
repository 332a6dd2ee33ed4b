#!/bin/bash
# usage: kres.sh file.hip [-Dflags...]  -> one line per kernel: VGPRs AGPRs scratch occupancy spills LDS (compile only, no GPU)
f=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/tumblr_emotions_amd/csrc -Wno-unused-result "$@" -c --cuda-device-only $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re, subprocess
rows, cur = [], None
for l in sys.stdin:
    m = re.search(r'remark: Function Name: (\S+)', l)
    if m:
        cur = [m.group(1)]
        rows.append(cur)
        continue
    m = re.search(r'remark: +(VGPRs|AGPRs|VGPRs Spill|ScratchSize|Occupancy|LDS Size)[^:]*: (\d+)', l)
    if m and cur is not None:
        cur.append('%s=%s' % (m.group(1).replace(' ', ''), m.group(2)))
for r in rows:
    name = subprocess.run(['c++filt', r[0]], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '')
    print('%-70s %s' % (name[:70], ' '.join(r[1:])))
"
