"""msdr_comm_* (include/msdr.h): the RCCL audio gather driven from C.  One GPU here, so the communicator has one rank -- which still
takes every step of the path: librccl is loaded by dlopen, ncclCommInitRank runs, the gather is queued on the communicator's own
stream behind an event of the context's stream, and the waits (device-side and host-side) release the slots.  The N > 1 data
movement itself is the same calls with peers; tests/test_dist_gloo.py covers the sharding and the double-buffered schedule on CPU."""
import numpy as np
import pytest

from gpuhelp import ctx, msdr  # noqa: F401

pytestmark = pytest.mark.gpu


def test_one_rank_gather_to_root_and_all_gather_overlapped_with_compute(ctx, orc, golden):
    uid = msdr.comm_unique_id()
    assert len(uid) == 128
    comm = msdr.Comm(ctx, uid, 0, 1)
    try:
        rng = np.random.default_rng(3)
        ch, n, blocks = 16, 2048, 6
        taps = golden["fir/taps_am102"]
        chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, taps, taps, mode=msdr.MODE_AM)
        x = rng.integers(-12000, 12001, (ch, blocks * n)).astype(np.int16)
        audio = [ctx.array((ch, n), np.int16) for _ in range(2)]
        gathered = [ctx.array((ch, n), np.int16) for _ in range(2)]
        got = np.empty_like(x)
        for k in range(blocks):
            slot = k % 2
            if k >= 2:                                        # the gather that last used this buffer pair: wait, then read its result
                comm.wait(slot, host_wait=True)
                got[:, (k - 2) * n:(k - 1) * n] = gathered[slot].download()
            dx = ctx.to_device(x[:, k * n:(k + 1) * n])
            chain.process(dx, audio[slot], n)
            comm.begin(slot, audio[slot].ptr, ch * n * 2, gathered[slot].ptr, 0 if k % 3 else -1)    # gather-to-root and all-gather
        for k in (blocks - 2, blocks - 1):
            comm.wait(k % 2, host_wait=(k == blocks - 1))     # one device-side wait, one host-side
            ctx.synchronize()
            got[:, k * n:(k + 1) * n] = gathered[k % 2].download()
        for c in (0, 7, ch - 1):
            assert np.array_equal(got[c], orc.chain_q15(x[c], msdr.MODE_AM, taps, taps))
        with pytest.raises(msdr.MsdrError):
            comm.begin(9, audio[0].ptr, 16, gathered[0].ptr, 0)
        with pytest.raises(msdr.MsdrError):
            comm.begin(0, audio[0].ptr, 16, gathered[0].ptr, 1)      # root outside the communicator
    finally:
        comm.close()
