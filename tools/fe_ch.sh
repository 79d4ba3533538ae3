#!/usr/bin/env bash
# tools/fe_ch.sh -- frontend_pipe4_kernel<CH>: channels per workgroup x batch shape on one box
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r02
for shape in "4096 262144" "8192 131072" "16384 65536" "32768 32768"; do
  set -- $shape
  for chw in 16 32 64; do
    MSDR_FRONTEND_PIPE_CH=$chw python bench.py --workload fe --channels $1 --samples $2 --no-cpu > gpurun_out/r02/ws.json 2>/dev/null && python -c "
import json
d=json.load(open('gpurun_out/r02/ws.json'))
print('fe per-workgroup $chw  ch $1 n $2 ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
  done
done
