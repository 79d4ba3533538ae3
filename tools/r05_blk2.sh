#!/usr/bin/env bash
# tools/r05_blk2.sh -- block cadence, the four fp32 shapes and the Q15 one, small and large batches (one line each)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MSDR_BENCH_NO_POWER=1
for args in "--workload c3" "--workload c4" "--workload c3 --channels 65536" "--workload c4 --channels 65536" "--workload c3 --arith q15" "--workload c3 --arith q15 --channels 65536" "--workload c5"; do
  tools/memguard.sh -m 24 -t 120 python3 bench.py $args --samples 128 --steps 300 --warmup 50 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; par=d.get('parity') or {}
print('$args', d['value'],'Msps', r.get('tick_us'),'us/tick kernel_ms', r['kernel_ms'], d['config']['kernel'][:40], 'grid', d['config']['grid'], 'parity', par.get('rel_rms_worst', par.get('mismatching_samples')))"
done
