#!/usr/bin/env bash
# tools/r05_nw.sh -- waves per workgroup of chain_mfb_kernel (MSDR_MB_NW), block cadence, four shapes
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MSDR_BENCH_NO_POWER=1
for nw in ${NWS:-2 4 6 8}; do
  for args in "--workload c4 --channels 65536" "--workload c3 --channels 65536" "--workload c4" "--workload c3"; do
    MSDR_MB_NW=$nw tools/memguard.sh -m 24 -t 120 python3 bench.py $args --samples 128 --steps 300 --warmup 50 --no-cpu --no-parity 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('nw $nw $args', d['value'],'Msps', round(d['ms_per_step']*1000,1),'us/tick kernel', round(r['kernel_ms']*1000,1), 'grid', d['config']['grid'])"
  done
done
