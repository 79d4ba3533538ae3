#!/usr/bin/env bash
# tools/pmc.sh <tag> <bench args...> -- SQ / GRBM counter passes (rocprofv3 --pmc only, no tracing domains) for the
# chain kernel of one bench.py command; prints per-launch means.  Run on the GPU box via gpurun.
set -u
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
export MSDR_BENCH_NO_POWER=1      # no rocm-smi child process under the profiler
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="$* --no-cpu --steps 2 --warmup 1"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_WAVES_EQ_64 SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/p$i" -- python3 bench.py $ARGS > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if ("chain_" in k and "kernel" in k) or "fir_f32" in k or "spectrum_rfft128" in k:
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(agg):
    print("%-32s mean per launch %18.1f  (%d launches)" % (k, agg[k][0] / agg[k][1], agg[k][1]))
PY
find "$OUT" -name "*.csv" -size +1M -delete
