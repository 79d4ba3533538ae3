import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib
from gpuhelp import msdr
import test_gpu_syncam as T
orc = orclib.Oracle(); ctx = msdr.Context(0)
rng = np.random.default_rng(1 + 4096)
i, q = T._iq(rng, 1, 4096)
i[0, 50:60], q[0, 50:60] = 32767, -32768
pll = msdr.Syncam(ctx, 1)
di, dq = ctx.to_device(i), ctx.to_device(q)
pll.process(di, dq, di, 4096)
got = di.download()[0]
s = orc.syncam_new(); want = orc.syncam_q15(s, i[0], q[0])
d = got.astype(int) - want
bad = np.nonzero(d)[0]
print("mismatches", len(bad), "first", bad[:10], "diffs", d[bad[:10]], "max", np.abs(d).max())
print("state gpu", pll.state(0), "orc", s.fil_out, s.omega2, s.phzerror)
