#!/usr/bin/env bash
# tools/wide_shapes.sh -- the channel-serial kernels (front end, Teensy biquad nodes) at the bench's shape and at shapes with more channels:
# one workgroup = 64 channels walks its stream alone, so 4096 channels keep 64 of the 256 CUs busy and the time is set by the stream length.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r02
for shape in "4096 262144" "8192 131072" "16384 65536" "32768 32768" "65536 16384"; do
  set -- $shape
  python bench.py --workload fe --channels $1 --samples $2 --no-cpu > gpurun_out/r02/ws.json 2>/dev/null && python -c "
import json
d=json.load(open('gpurun_out/r02/ws.json'))
print('fe      ch $1 n $2 ms', d['ms_per_step'], 'Msamples/s', d['value'], 'frac', d['roofline']['frac'])"
  python bench.py --workload c3 --arith q15 --channels $1 --samples $2 --no-cpu --no-parity > gpurun_out/r02/ws.json 2>/dev/null && python -c "
import json
d=json.load(open('gpurun_out/r02/ws.json'))
print('c3 q15  ch $1 n $2 ms', d['ms_per_step'], 'Msamples/s', d['value'], 'kernel_ms', d['roofline'].get('kernel_ms'))"
done
