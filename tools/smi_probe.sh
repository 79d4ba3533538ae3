#!/usr/bin/env bash
# tools/smi_probe.sh <tag> [ENV=VAL ...] -- <bench args>: samples rocm-smi clocks / power while the bench loops
TAG=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
if [ $# -eq 0 ]; then echo "usage: $0 <tag> [ENV=VAL ...] -- <bench args>" >&2; exit 2; fi
shift
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r03p
( for e in "${ENVS[@]}"; do export "$e"; done; python3 bench.py "$@" --no-cpu --no-parity --steps ${STEPS:-6000} --warmup 5 > gpurun_out/r03p/smi_$TAG.json 2>/dev/null ) &
BP=$!
sleep ${WAIT:-5}
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo
  sleep 0.7
done
wait $BP
python3 -c "
import json; d=json.load(open('gpurun_out/r03p/smi_$TAG.json')); print('$TAG', 'ms', d['ms_per_step'])"
