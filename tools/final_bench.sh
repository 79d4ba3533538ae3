#!/usr/bin/env bash
# tools/final_bench.sh -- every bench line of DESIGN.md section 6 on one box (run through gpurun); JSON lines under gpurun_out/final/.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/final; mkdir -p $O
run() { name=$1; shift; python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%-22s %10.1f Ms/s  step %8.4f ms  kernel %s ms  frac %.4f  %s  cpu %s" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("kernel_ms"), r["frac"],
          d.get("parity"), (d.get("cpu_baseline") or {}).get("value")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run c2 --workload c2
run c3 --workload c3
run c4 --workload c4
run c5 --workload c5
run c2_nomfma --workload c2 --no-mfma
run c3_nomfma --workload c3 --no-mfma
run q15_c3 --arith q15 --workload c3
run q15_c4 --arith q15 --workload c4
run q15_c5 --arith q15 --workload c5
run q15_c2_demod --arith q15 --workload c2 --stages 0
run q15_c3_demod --arith q15 --workload c3 --stages 0
run fir_f32 --workload fir
run fir_q15 --workload fir --arith q15
run fe --workload fe
run spec --workload spec
