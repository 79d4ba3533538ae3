#!/usr/bin/env bash
# tools/tq_ch.sh -- biquad_teensy_pipe4_kernel<2, CH>: channels per workgroup x batch shape on one box (Q15 whole step, c3 / c5 taps)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r02
for cfg in "c3 4096 262144" "c3 8192 131072" "c3 16384 65536" "c5 256 1048576"; do
  set -- $cfg
  for chw in 16 32 64; do
    MSDR_BIQUAD_PIPE_CH=$chw python bench.py --workload $1 --arith q15 --channels $2 --samples $3 --no-cpu --no-parity > gpurun_out/r02/ws.json 2>/dev/null && python -c "
import json
d=json.load(open('gpurun_out/r02/ws.json'))
print('$1 q15 per-workgroup $chw  ch $2 n $3 ms', d['ms_per_step'], 'Msamples/s', d['value'])"
  done
done
