"""`python bench.py --gpus N` starts its own ranks (bench.launch_ranks: a CHILD torch.distributed.run job, never an exec of a process
that has touched the GPU).  Here, without a GPU, the hop itself is checked: both ranks start, see WORLD_SIZE = 2 and stop at the
library's "needs a GPU" gate; the parent relays the failure as its exit code.  The two-rank run proper is tests/test_gpu_bench_ranks.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_starts_two_ranks_as_a_child_job():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU box: the real two-rank run is test_gpu_bench_ranks.py")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") == 2, r.stderr[-2000:]      # one per rank: the launcher really started two
    assert "rank      : 1 (local_rank: 1)" in r.stderr


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
