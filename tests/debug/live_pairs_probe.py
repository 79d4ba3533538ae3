"""tests/debug/live_pairs_probe.py -- two live updates with no samples in between (tests/debug/fuzz_live.py seed 5202 case 44155): every ordered
pair of {taps, biquad, mode, osc} on a small fp32 chain with one channel of every mode, against the oracle's continuation."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpuhelp import msdr, rel_rms  # noqa: E402
import orclib  # noqa: E402
from test_gpu_chain import _f32_biquads, _hilbert_pair, _q15_nco  # noqa: E402

orc = orclib.Oracle()
ctx = msdr.Context(0)
rng = np.random.default_rng(1)
ntaps = 100
modes = np.array([orclib.AM, orclib.LSB, orclib.USB, orclib.CW], np.int32)
bad = 0
for mixer in (0, 1):
    for first, second in itertools.product(("taps", "biquad", "mode", "osc"), repeat=2):
        if not mixer and "osc" in (first, second):
            continue
        hi, hq = _hilbert_pair(ntaps)
        bq = _f32_biquads(orc, 2)
        oi, oq = _q15_nco(4, 1) if mixer else (np.array([0, 1, 0, -1], np.float32), np.array([1, 0, -1, 0], np.float32))
        m = modes.copy()
        chain = msdr.Chain(ctx, msdr.ARITH_F32, 4, hi, hq, mixer=mixer, modes=m, osc_i=oi if mixer else None, osc_q=oq if mixer else None, biquad_coeffs=bq)
        states = {c: {} for c in range(4)}
        def run(n):
            global bad
            x = rng.integers(-20000, 20001, (4, n)).astype(np.int16)
            dx, dy = ctx.to_device(x), ctx.array((4, n), np.float32)
            chain.process(dx, dy, n)
            got = dy.download()
            errs = []
            for c in range(4):
                want = orc.chain_f32(x[c], int(m[c]), hi, hq, oi, oq, bq, state=states[c])
                errs.append(rel_rms(got[c], want))
            return errs
        run(1024)
        for op in (first, second):
            if op == "taps":
                hi, hq = _hilbert_pair(ntaps, fc=1500.0 + 300 * rng.random(), bw=2400.0)
                chain.set_taps(0, hi, hq)
            elif op == "biquad":
                bq = bq.copy()
                # (a well-conditioned section: the cascade stays inside the chain kernel, folded into the SSB tables)
                c30 = orc.biquad_design(orclib.BQ_LOWPASS, np.float32(3500.0 + 2500 * rng.random()), 0.6 + 0.6 * rng.random()).astype(np.float64) / 2 ** 30
                bq[1] = [c30[0], c30[1], c30[2], -c30[3], -c30[4]]
                chain.set_biquad_coeffs(bq)
            elif op == "mode":
                m[1], m[2] = m[2], m[1]
                chain.set_mode(1, int(m[1]), 0); chain.set_mode(2, int(m[2]), 0)
            elif op == "osc":
                oi, oq = _q15_nco(8, 1) if oi.size and rel_rms(oi[:8], _q15_nco(8, 1)[0][:8]) > 0.1 else _q15_nco(4, 1)
                chain.set_osc(oi, oq)
        e1 = run(128)
        e2 = run(1024)
        worst = max(e1 + e2)
        flag = "BAD" if worst > 1e-5 else "ok"
        bad += worst > 1e-5
        print("%-3s mixer %d  %-6s then %-6s  first block %s   next %s   %s" % (flag, mixer, first, second, " ".join("%.1e" % e for e in e1), " ".join("%.1e" % e for e in e2), chain.info()["kernel"][:40]), flush=True)
        chain.close()
print("live_pairs_probe done: %d pairs beyond 1e-5" % bad)
