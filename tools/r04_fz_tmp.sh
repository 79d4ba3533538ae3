cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r04f
timeout -k 10 400 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "fp32 contract|passed|failed|FAILED" | cut -c1-300
timeout -k 10 330 python tests/debug/fuzz_f32_truth.py 200 4106 > gpurun_out/r04f/fuzz_f32_truth_4106c.txt 2>&1; tail -1 gpurun_out/r04f/fuzz_f32_truth_4106c.txt | cut -c1-300
timeout -k 10 330 python tests/debug/fuzz_stage_df1.py 150 4107 > gpurun_out/r04f/fuzz_stage_df1_4107c.txt 2>&1; tail -1 gpurun_out/r04f/fuzz_stage_df1_4107c.txt | cut -c1-300
