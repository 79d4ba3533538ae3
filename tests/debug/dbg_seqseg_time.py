"""Time of the CMSIS-order fallback on one long stream, with and without time segments (MSDR_BIQUAD_SEQ_NO_SEGMENTS=1)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr  # noqa: E402
import orclib  # noqa: E402
orc = orclib.Oracle()
ctx = msdr.Context(0)
c = orc.biquad_design(orclib.BQ_HIGHPASS, np.float32(300.0), 0.707).astype(np.float64) / 2 ** 30
coeffs = np.array([[c[0], c[1], c[2], -c[3], -c[4]]] * 2, np.float32)
for ch, n in ((1, 1 << 24), (16, 1 << 22), (1024, 1 << 16)):
    x = ctx.to_device((0.5 + 0.3 * np.random.default_rng(1).standard_normal((ch, n))).astype(np.float32))
    y = ctx.array((ch, n), np.float32)
    bq = msdr.BiquadDf1F32(ctx, coeffs, ch)
    bq.process(x, y, n); ctx.synchronize()
    t0 = time.perf_counter()
    bq.process(x, y, n); ctx.synchronize()
    dt = time.perf_counter() - t0
    print("segments %s: %5d channel(s) x %9d samples, two 300 Hz high-pass sections in CMSIS order: %8.2f ms = %7.1f Msamples/s"
          % ("off" if os.environ.get("MSDR_BIQUAD_SEQ_NO_SEGMENTS") else "on ", ch, n, dt * 1e3, ch * n / dt / 1e6), flush=True)
    bq.close()
