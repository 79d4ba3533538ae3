"""bench.py with N > 1 ranks: the control flow the driver's multi-GPU run takes (launcher hop, per-rank shards, max-over-ranks timing,
records that exist on rank 0 only, the gather section), rehearsed with two ranks on the ONE card of the test box
(MSDR_BENCH_REHEARSAL=1: gloo collectives on host tensors; a dry run, not a measurement)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_rehearsal_prints_one_complete_line():
    # `python bench.py --gpus 2` on its own: bench.py starts the two ranks itself (a child torch.distributed.run job, launch_ranks)
    env = dict(os.environ, MSDR_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                       # rank 0 prints, nobody else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and len(d["rank_devices"]) == 2 and d["scaling"] == "weak"
    assert len(lines[0]) < 6000                                  # the driver keeps a log tail: the whole line must fit in it
    recs = d["roofline"]["records"]                              # compact summaries of the sub-records, inside the top-level roofline
    assert {"fir", "fir512", "c2", "c4", "c5", "q15_c3", "c3_i16", "c3_b128", "c4_b128", "q15_c3_b128", "q15_c1_b128", "update_all"} <= set(recs)
    assert d["library_rev"]
    assert d["roofline"]["frac"] > 0 and d["parity"] is not None
    for name, rec in recs.items():
        assert rec["frac"] > 0 and rec["kernel_ms"] > 0 and rec["ms_per_step"] >= rec["kernel_ms"] * 0.98, name
        if name not in ("fir", "fir512"):                        # (the FIR stage's parity needs the CPU leg, off here)
            assert any(rec.get(k) is not None for k in ("parity", "parity_mismatches", "parity_max_lsb")), name
    assert d["gather"]["all_gather_ms"] > 0 and d["gather"]["gather_to_root_ms"] > 0
    assert "rehearsal" in d
    full = json.load(open(os.path.join(ROOT, "bench_full.json")))           # the full record beside the script
    assert set(full["also"]) == set(recs)
    assert {"all_gather", "gather_to_root", "overlapped", "c_abi"} <= set(full["gather"])
    for name, rec in full["also"].items():
        if name not in ("fir", "fir512"):
            assert rec["parity"]["windows"], name
