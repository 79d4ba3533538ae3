"""The N > 1 path on CPU: two processes, gloo backend, 127.0.0.1.  Each rank demodulates its channel
shard (the oracle stands in for the HIP chain, which needs a GPU) and the audio is gathered exactly as
bench.py / a multi-GPU caller does through minimal-sdr_amd/python/msdr_dist.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimal-sdr_amd", "python"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import msdr_dist
    import orclib
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = orclib.Oracle()
    start, count = msdr_dist.channel_shard(total, world, rank)
    taps = orc.calc_fir_coeffs(102, 2800)[:102]
    rng = np.random.default_rng(1234)
    x_all = rng.integers(-12000, 12001, (total, n)).astype(np.int16)          # same seed everywhere; a rank only touches its rows
    modes = np.array([orclib.AM, orclib.LSB, orclib.USB, orclib.AM, orclib.CW][:total] + [orclib.AM] * max(0, total - 5), np.int32)
    local = np.stack([orc.chain_q15(x_all[c], modes[c], taps, taps) for c in range(start, start + count)]) if count else np.zeros((0, n), np.int16)
    full = msdr_dist.gather_audio(torch.from_numpy(local), total)
    slowest = msdr_dist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    # gather to one rank only (the play-out host): root 1 here, so that rank 0 is a pure sender
    at_root = msdr_dist.gather_audio_root(torch.from_numpy(local), total, root=1)
    root_ok = torch.tensor([1.0 if (at_root is None) == (rank != 1) else 0.0])
    if rank == 1:
        root_ok[0] = float(torch.equal(at_root, full))
    dist.all_reduce(root_ok, op=dist.ReduceOp.MIN)
    # double-buffered schedule: block k is gathered while block k + 1 is demodulated; fp32 audio this time (4-byte samples)
    blocks, nb = 5, n // 2
    state = [dict() for _ in range(count)]
    og = msdr_dist.OverlappedGather(total, count, nb, torch.float32, torch.device("cpu"), root=0)
    og_all = msdr_dist.OverlappedGather(total, count, nb, torch.float32, torch.device("cpu"), root=None)
    cos4, sin4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)
    hf = taps.astype(np.float32) / 32768.0
    got_blocks = []
    for k in range(blocks):
        buf, buf2 = og.buffer(k), og_all.buffer(k)
        if k >= 2 and rank == 0:
            got_blocks.append(og.result(k).clone())          # block k - 2, complete by now (buffer() waited for it)
        xb = x_all[start:start + count, (k % 2) * nb:(k % 2 + 1) * nb]
        for c in range(count):
            buf[c] = torch.from_numpy(orc.chain_f32(xb[c], modes[start + c], hf, hf, sin4, cos4, state=state[c]))
        buf2.copy_(buf)
        og.submit(k)
        og_all.submit(k)
    og.finish(); og_all.finish()
    ov_ok = torch.tensor([1.0])
    last_all = og_all.result(blocks - 1)
    if rank == 0:
        got_blocks += [og.result(blocks - 2).clone(), og.result(blocks - 1).clone()]
        st = [dict() for _ in range(total)]
        for k in range(blocks):
            want_k = np.stack([orc.chain_f32(x_all[c, (k % 2) * nb:(k % 2 + 1) * nb], modes[c], hf, hf, sin4, cos4, state=st[c]) for c in range(total)])
            if not np.array_equal(got_blocks[k].numpy(), want_k):
                ov_ok[0] = 0.0
        if not torch.equal(last_all, got_blocks[-1]):
            ov_ok[0] = 0.0
    else:
        assert og.result(blocks - 1) is None
    dist.all_reduce(ov_ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        want = np.stack([orc.chain_q15(x_all[c], modes[c], taps, taps) for c in range(total)])
        q.put((bool(np.array_equal(full.numpy(), want)) and bool(root_ok.item() == 1.0) and bool(ov_ok.item() == 1.0), slowest, (start, count)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 4])
def test_two_rank_channel_shard_and_audio_gather(total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, 256, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, slowest, shard0 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
    assert slowest == 2.0
    assert shard0 == (0, 3 if total == 5 else 2)


def test_channel_shard_covers_everything():
    import msdr_dist
    for total in (1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [msdr_dist.channel_shard(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
