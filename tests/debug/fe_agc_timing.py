"""Front end step time with AGC() on and off (GPU box, from the repository root): is the one-lane-per-channel AGC wave the long pole of a step?
    gpurun -- python tests/debug/fe_agc_timing.py [channels] [samples]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr  # noqa: E402
import torch  # noqa: E402

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
ctx = msdr.Context(0)
x = (32768 + torch.randint(-6000, 6000, (ch, n), device="cuda", dtype=torch.int32)).to(torch.int16)
y = torch.empty_like(x)
for agc in (1, 0, 1, 0):
    fe = msdr.Frontend(ctx, ch)
    fe.prime(np.uint16(32768))
    fe.set_agc(agc)
    for _ in range(3):
        fe.update(x.data_ptr(), y.data_ptr(), n)
    ctx.synchronize() if hasattr(ctx, "synchronize") else torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fe.update(x.data_ptr(), y.data_ptr(), n)
    ctx.synchronize() if hasattr(ctx, "synchronize") else None
    torch.cuda.synchronize()
    print("channels %d samples %d AGC %d: %.3f ms per step" % (ch, n, agc, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
    fe.close()
