#!/usr/bin/env bash
# tools/ab_tq_run.sh -- the FIR stage with runs of 1 / 2 / 4 tiles per draw (MSDR_TQ_RUN_SHIFT), same box, alternating rounds
set -u
O=gpurun_out/tq_run; mkdir -p $O
for round in 1 2 3; do
 for rs in 0 1 2 3; do
  MSDR_TQ_RUN_SHIFT=$rs python bench.py --workload fir --no-cpu > $O/fir_rs${rs}_r${round}.json 2> $O/err.txt
  python - <<PY
import json
d=json.load(open("$O/fir_rs${rs}_r${round}.json"))
r=d["roofline"]
print("round $round run_shift $rs", d["config"].get("kernel"), "kernel_ms", r["kernel_ms"], "frac", r["frac"], "sclk", r.get("sclk_mhz"), "W", r.get("power_w"), "parity", d.get("parity", {}).get("rel_rms_worst"))
PY
 done
done
