#!/usr/bin/env bash
# tools/pmc_coexec.sh <workload> -- how much of the matrix pipe's busy time overlaps vector issue (one rocprofv3 --pmc pass, no tracing)
set -u
W=$1
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
export MSDR_BENCH_NO_POWER=1      # no rocm-smi child process under the profiler
OUT=gpurun_out/pmc_coexec_$W
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for SET in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" \
           "SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/p$i" -- python3 bench.py --workload $W --no-cpu --no-parity --steps 2 --warmup 1 > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if ("chain_" in k and "kernel" in k) or "fir_f32" in k:
            a = agg[(k[:60], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(agg):
    print("%-62s %-32s mean per launch %18.1f  (%d launches)" % (k[0], k[1], agg[k][0] / agg[k][1], agg[k][1]))
PY
find "$OUT" -name "*.csv" -size +1M -delete
