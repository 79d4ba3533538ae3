"""Prints the conditioning figures (kappa, fp32 noise, decision) the library computes for the cascades the tests use:
    MSDR_DEBUG_CONDITION=1 python tests/debug/dbg_condition.py      (GPU box: creating an instance needs a context)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MSDR_DEBUG_CONDITION"] = "1"
from gpuhelp import msdr  # noqa: E402
import orclib  # noqa: E402
orc = orclib.Oracle()
ctx = msdr.Context(0)
CORR = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0


def rows(specs):
    out = []
    for kind, f, q in specs:
        c = orc.biquad_design(kind, np.float32(f), q).astype(np.float64) / 2 ** 30
        out.append([c[0], c[1], c[2], -c[3], -c[4]])
    return np.array(out, np.float32)


LP, NT, HP = orclib.BQ_LOWPASS, orclib.BQ_NOTCH, orclib.BQ_HIGHPASS
for name, specs in [("reference: LP 0.54 + notch 15", [(LP, 5400 * CORR, 0.54), (NT, 3000 * CORR, 15.0)]),
                    ("test cascade, 3 sections", [(LP, 5400 * CORR, 0.54), (NT, 3000 * CORR, 15.0), (LP, 5400 * CORR, 0.54)]),
                    ("test cascade, 4 sections", [(LP, 5400 * CORR, 0.54), (NT, 3000 * CORR, 15.0), (LP, 5400 * CORR, 0.54), (LP, 5400 * CORR, 1.3)]),
                    ("Linkwitz-Riley LP set (.ino:393-399)", [(LP, 5400 * CORR, q) for q in (0.54, 1.3, 0.54, 1.3)]),
                    ("one 300 Hz high-pass", [(HP, 300.0, 0.707)]),
                    ("notch Q 20 at 300 Hz + LP", [(NT, 300.0, 20.0), (LP, 5000.0, 0.7)])]:
    print(name, flush=True)
    msdr.BiquadDf1F32(ctx, rows(specs), 1).close()
