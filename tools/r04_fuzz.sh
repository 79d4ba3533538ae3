#!/usr/bin/env bash
# tools/r04_fuzz.sh <part> -- the randomised differential tests of tests/debug/ at the round's final revision (run through gpurun; every
# fuzzer stays below five minutes so that the call keeps producing output).  Output: gpurun_out/r04f/*.txt, tail lines = the summaries.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r04f; mkdir -p $O
run() { name=$1; shift; timeout -k 10 330 python tests/debug/$name.py "$@" > $O/${name}_$2.txt 2>&1; echo "$name $*: rc $?"; tail -2 $O/${name}_$2.txt | cut -c1-400; }
case ${1:-a} in
  a) run fuzz_kernels 280 4101; run fuzz_live 240 4102; run fuzz_fir_f32 200 4103 ;;
  b) run fuzz_retune 150 4104; run fuzz_retune_q15 150 4105; run fuzz_f32_truth 240 4106; run fuzz_stage_df1 120 4107 ;;
  c) run fuzz_pll_anr 120 4108; run fuzz_pll_anr_f32 120 4109; run fuzz_frontend 120 4110; run fuzz_kernels 280 4111 ;;
  d) run fuzz_kernels 300 4201; run fuzz_live 300 4202; run fuzz_f32_truth 300 4206 ;;     # a second pass with fresh seeds at the round's last revision
esac
