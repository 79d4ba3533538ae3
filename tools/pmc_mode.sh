#!/usr/bin/env bash
# tools/pmc_mode.sh <workload> <MSDR_DBG mode> -- matrix-pipe busy share and effective clock of an ablation mode of the experiment library
set -u
W=$1; M=$2
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
export MSDR_DBG=$M MSDR_LIB=$PWD/minimal-sdr_amd/lib_ab/libmsdr_prio.so
OUT=gpurun_out/pmc_mode_$M
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d "$OUT/p1" -- python3 bench.py --workload $W --no-cpu --no-parity --steps 20 --warmup 10 > "$OUT/p1.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if ("chain_" in k and "kernel" in k):
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
m = {k: v[0] / v[1] for k, v in agg.items()}
ms = None
for line in open(os.path.join(out, "p1.log")):
    if line.startswith("{"):
        ms = json.loads(line)["roofline"]["kernel_ms"]
cyc = m["GRBM_GUI_ACTIVE"] / 8
print("mode %s: kernel_ms %.4f  cycles/XCD %.0f  effective clock %.3f GHz  mfma busy %.3f  valu active %.3f  coexec/mfma %.3f  mfma/tile-ish %.0f valu %.0f" % (
    os.environ["MSDR_DBG"], ms, cyc, cyc / ms / 1e6, m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, 4 * m["SQ_ACTIVE_INST_VALU"] / 1024 / cyc,
    m["SQ_VALU_MFMA_COEXEC_CYCLES"] / m["SQ_VALU_MFMA_BUSY_CYCLES"], m["SQ_INSTS_MFMA"], m["SQ_INSTS_VALU"]))
PY
find "$OUT" -name "*.csv" -size +1M -delete
