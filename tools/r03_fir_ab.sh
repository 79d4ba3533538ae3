#!/usr/bin/env bash
# tools/r03_fir_ab.sh -- fir_f32tq_kernel (tile queue) against fir_f32tr_kernel (one stream per wave) on one box, alternating; then the
# number of fronts.  Writes gpurun_out/r03/fir_ab.txt
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03/fir_ab.txt
: > $OUT
run() {   # label, env...
  local label=$1; shift
  env "$@" python bench.py --workload fir --no-cpu --steps 20 --warmup 3 > gpurun_out/r03/fir_ab_x.json 2> gpurun_out/r03/fir_ab_x.err || { echo "$label FAILED" >> $OUT; tail -3 gpurun_out/r03/fir_ab_x.err >> $OUT; return; }
  python -c "
import json
d=json.load(open('gpurun_out/r03/fir_ab_x.json'))
print('$label', d['config']['kernel'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])" >> $OUT
}
for rep in 1 2; do
  run "stream-per-wave" MSDR_FIR_NO_TQ=1
  run "queue-32-fronts" MSDR_FIR_TQ_FRONTS=32
done
for f in 8 16 24 40 64; do run "queue-$f-fronts" MSDR_FIR_TQ_FRONTS=$f; done
cat $OUT
