#!/usr/bin/env bash
# tools/exp.sh <workload> "<args>" ... : run bench.py variants (no CPU leg) and print value + kernel time
W=$1; shift
for a in "$@"; do
  timeout 300 python bench.py --workload $W --steps 5 --warmup 1 --no-cpu $a 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s %-40s %10.1f Ms/s  kernel %8.4f ms  %s' % ('$W', '$a', j['value'], j['roofline']['kernel_ms'], j['config']['kernel']))"
done
