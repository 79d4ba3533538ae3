#!/usr/bin/env bash
# tools/fir_tq_power.sh -- every ablation of tools/probes/fir_tq_bench looped for a few seconds while rocm-smi samples the shader
# clock and the package power: which parts of fir_f32tq_kernel run at the 1400 W cap, and at what clock.  Output: gpurun_out/fir_tq_power.txt
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
OUT=gpurun_out/fir_tq_power.txt
: > $OUT
for v in ${VARIANTS:-0 1 2 3 8 24 10}; do
  ./tools/probes/fir_tq_bench 32 $v ${SECS:-7} > gpurun_out/fir_tq_power_$v.json 2>&1 &
  BP=$!
  sleep 3
  for i in 1 2 3; do
    echo -n "variant $v: " >> $OUT
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $OUT; echo >> $OUT
    sleep 0.8
  done
  wait $BP
  cat gpurun_out/fir_tq_power_$v.json >> $OUT
done
[ -n "${VARIANTS:-}" ] || ./tools/probes/fir_tq_bench 32 > gpurun_out/fir_tq_fastfir.jsonl 2>&1
cat $OUT
cat gpurun_out/fir_tq_fastfir.jsonl
