#!/usr/bin/env bash
# tools/ab_libs.sh <out-tag> <libA> <libB> -- <workload> [<workload> ...]: same-box A/B of two builds of the library, three alternating
# rounds per workload (bench.py --workload X --no-cpu); prints kernel_ms, frac, sclk, parity per run.  Output: gpurun_out/ab_<tag>.txt
TAG=$1; A=$2; B=$3; shift 3; [ "$1" = "--" ] && shift
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
OUT=gpurun_out/ab_$TAG.txt
: > $OUT
for wl in "$@"; do
  for rep in 1 2 3; do for lib in "$A" "$B"; do
    MSDR_LIB=$PWD/$lib python bench.py --workload $wl --no-cpu ${AB_ARGS:-} > gpurun_out/ab_x.json 2>/dev/null
    python - "$wl" "$lib" >> $OUT <<'PY'
import json, sys
try:
    d = json.load(open('gpurun_out/ab_x.json'))
    r = d['roofline']; p = d.get('parity', {})
    print('%-4s %-40s step_ms %.4f kernel_ms %.4f frac %.4f sclk %s W %s parity %s' % (sys.argv[1], sys.argv[2].split('/')[-1], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('sclk_mhz'), r.get('power_w'),
          p.get('rel_rms_worst', p.get('mismatching_samples'))))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e)
PY
  done; done
done
cat $OUT
