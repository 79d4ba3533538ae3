"""tests/debug/firq_block_time.py -- wall time per msdr_fir_q15_process call at the reference's block length (no events: calls queued back to back)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpuhelp import msdr  # noqa: E402

ctx = msdr.Context(0)
rng = np.random.default_rng(1)
for ch in (1, 64, 4096, 16384):
    for ntaps in (102, 256):
        taps = rng.integers(-2500, 2501, ntaps).astype(np.int16)
        x = rng.integers(-20000, 20001, (ch, 128)).astype(np.int16)
        fir = msdr.FirQ15(ctx, taps, ch)
        dx, dy = ctx.to_device(x), ctx.array((ch, 128), np.int16)
        for _ in range(200):
            fir.process(dx, dy, 128)
        ctx.synchronize()
        t0 = time.perf_counter()
        K = 2000
        for _ in range(K):
            fir.process(dx, dy, 128)
        ctx.synchronize()
        us = (time.perf_counter() - t0) / K * 1e6
        print("channels %6d taps %3d: %.2f us per call = %.1f Gsamples/s  (MSDR_NO_BLOCK=%s)" % (ch, ntaps, us, ch * 128 / us / 1e3, os.environ.get("MSDR_NO_BLOCK", "")), flush=True)
        fir.close()
