#!/usr/bin/env bash
# tools/clock_probe.sh <tag> [ENV=VAL ...] -- <bench args>: GRBM_GUI_ACTIVE cycles (counter pass) and traced duration (trace pass) of the chain kernel -> effective shader clock
TAG=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
if [ $# -eq 0 ]; then echo "usage: $0 <tag> [ENV=VAL ...] -- <bench args>" >&2; exit 2; fi
shift
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
OUT=gpurun_out/clk_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
for e in "${ENVS[@]}"; do export "$e"; done
# two passes: the counter pass is never combined with a tracing domain (tools/profile.sh), durations come from the trace pass
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 bench.py "$@" --no-cpu --no-parity --steps 3 --warmup 1 > "$OUT/log_trace.txt" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -- python3 bench.py "$@" --no-cpu --no-parity --steps 3 --warmup 1 > "$OUT/log_pmc.txt" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys
out, tag = sys.argv[1], sys.argv[2]
cyc = {}; dur = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "chain_" in k or "fir_f32" in k:
            cyc.setdefault(k[:60], []).append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "chain_" in k or "fir_f32" in k:
            dur.setdefault(k[:60], []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in cyc:
    c = sum(cyc[k]) / len(cyc[k]); d = sum(dur.get(k, [0])) / max(1, len(dur.get(k, [])))
    print(tag, k, "cycles/launch(all SE sum?) %.0f" % c, "dur_us %.1f" % (d / 1e3), "GHz(if per-XCD sum of 8) %.3f" % (c / 8 / max(d, 1)))
PY
