#!/usr/bin/env bash
# tools/r05_fuzz.sh <part> -- the randomised differential tests of tests/debug/ at the round's final revision (run through gpurun; every
# fuzzer stays below five minutes so that the call keeps producing output).  Output: gpurun_out/r05f/*.txt, tail lines = the summaries.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r05f; mkdir -p $O
run() { name=$1; shift; tools/memguard.sh -m 24 -t 330 python3 tests/debug/$name.py "$@" > $O/${name}_$2.txt 2>&1; echo "$name $*: rc $?"; tail -2 $O/${name}_$2.txt | cut -c1-400; }
case ${1:-a} in
  a) run fuzz_kernels 280 5101; run fuzz_live 240 5102; run fuzz_fir_f32 200 5103 ;;
  b) run fuzz_retune 150 5104; run fuzz_retune_q15 150 5105; run fuzz_f32_truth 240 5106; run fuzz_stage_df1 120 5107 ;;
  c) run fuzz_pll_anr 120 5108; run fuzz_pll_anr_f32 120 5109; run fuzz_frontend 120 5110; run fuzz_kernels 280 5111 ;;
  d) run fuzz_kernels 300 5201; run fuzz_live 300 5202; run fuzz_f32_truth 300 5206 ;;     # a second pass with fresh seeds at the round's last revision
esac
# round 5: the same live-update fuzzer at the reference's cadence (FUZZ_BLOCK=1: calls of 32 .. 512 samples: the block kernels)
if [ "${1:-a}" = "e" ]; then
  export FUZZ_BLOCK=1
  run fuzz_live 300 ${SEED1:-5301}; run fuzz_live 300 ${SEED2:-5302}
fi
# after the round's two live-update fixes: the seed that found the second defect again, and fresh ones (plain and block cadence)
if [ "${1:-a}" = "f" ]; then
  run fuzz_live 300 5202; run fuzz_live 300 5203
  export FUZZ_BLOCK=1
  run fuzz_live 300 5331
fi
if [ "${1:-a}" = "g" ]; then      # the FIR stage's self-resetting tile queue, the kernels and the retune paths once more on the last library
  run fuzz_fir_f32 200 5113; run fuzz_kernels 280 5211; run fuzz_retune 150 5204
fi
if [ "${1:-a}" = "h" ]; then      # the retune / truth / stage fuzzers once more on the round's last library
  run fuzz_retune 150 5214; run fuzz_retune_q15 150 5215; run fuzz_f32_truth 240 5216; run fuzz_stage_df1 100 5217
fi
