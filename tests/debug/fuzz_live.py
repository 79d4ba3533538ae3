"""Live updates under a running stream (msdr_chain_set_taps / set_osc / set_biquad_coeffs / set_node_coefficients, mixed with
msdr_chain_set_mode and msdr_chain_init_fir), random plans on random chains of both arithmetics, every case reproducible on its own
(run from the repository root on a GPU box):
    gpurun -- python tests/debug/fuzz_live.py [seconds] [seed] [only_case]
The oracle keeps its state and is handed the changed arrays, as the reference's callers do (UI.cpp:332-345, Minimal-SDR.ino:221-223, :356,
freq_conv.h:33-34).  Q15: bit-exact.  fp32: 1e-5 per channel and per stretch between two calls -- except where the float64 criterion of
tests/test_gpu_f32_contract.py excuses a resonant cascade.  Each case draws from default_rng([seed, case])."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr, rel_rms  # noqa: E402  (imports torch first)
import orclib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
SKIP = set(os.environ.get("FUZZ_SKIP_OPS", "").split(","))      # debugging a finding: leave operations of a kind out of the plan (same random draws)
orc = orclib.Oracle()
ctx = msdr.Context(0)
B = 128
CORR = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0


def f32_section(rng):
    kind = int(rng.choice([orclib.BQ_LOWPASS, orclib.BQ_NOTCH, orclib.BQ_LOWPASS]))
    c = orc.biquad_design(kind, np.float32(rng.uniform(1500, 8000) * CORR), float(rng.uniform(0.5, 4.0) if kind == orclib.BQ_LOWPASS else rng.uniform(3, 15))).astype(np.float64) / 2 ** 30
    return [c[0], c[1], c[2], -c[3], -c[4]]


def q15_section(rng):
    kind = int(rng.choice([orclib.BQ_LOWPASS, orclib.BQ_NOTCH, orclib.BQ_HIGHPASS]))
    return orc.biquad_design(kind, np.float32(rng.uniform(300, 9000) * CORR), float(rng.uniform(0.5, 15)))


def table(rng, q15):
    k = np.arange(B)
    kind = int(rng.integers(0, 4))
    if kind == 3:          # an unstructured table
        s, c = rng.uniform(-1, 1, B), rng.uniform(-1, 1, B)
    else:
        P = int(rng.choice([4, 8, 32, 128]))
        cyc = int(rng.integers(1, 6)) if P == 128 else 1
        s, c = np.sin(2 * np.pi * cyc * k / P), np.cos(2 * np.pi * cyc * k / P)
    si, ci = np.round(32767 * s).astype(np.int16), np.round(32767 * c).astype(np.int16)
    return (si, ci) if q15 else ((si / 32768.0).astype(np.float32), (ci / 32768.0).astype(np.float32))


t_end = time.time() + budget
case = bad = excused = 0
ops_seen = {}
kernels_seen = {}
while time.time() < t_end:
    case += 1
    if only >= 0:
        if case > 1:
            break
        case = only
    rng = np.random.default_rng([seed, case])
    q15 = bool(rng.integers(0, 2))
    ntaps = int(rng.integers(1, 130)) * 2 if q15 else int(rng.integers(2, 270))
    ch = int(rng.choice([1, 3, 66, 16, 48] if os.environ.get("FUZZ_BLOCK") else [1, 3, 66]))      # (multiples of 16 at block cadence: the sub-slab node kernel)
    nsets = int(rng.integers(1, 3))

    def taps():
        if q15:
            return rng.integers(-int(rng.choice([60, 4000])), int(rng.choice([60, 4000])) + 1, ntaps).astype(np.int16)
        return (rng.standard_normal(ntaps) / np.sqrt(ntaps) * rng.choice([1.0, 0.05])).astype(np.float32)
    sets_i = [taps() for _ in range(nsets)]
    sets_q = [s.copy() if rng.integers(0, 2) else taps() for s in sets_i]
    mixer = int(rng.integers(0, 2))
    oi, oq = table(rng, q15) if mixer else (None, None)
    modes = rng.choice([orclib.AM, orclib.LSB, orclib.USB, orclib.CW], ch).astype(np.int32)
    tapsets = rng.integers(0, nsets, ch).astype(np.int32)
    stages = int(rng.integers(0, 3))
    if q15:
        nodes = [[q15_section(rng)] for _ in range(stages)]
        chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, sets_i, sets_q, mixer=mixer, modes=modes, tapsets=tapsets, osc_i=oi, osc_q=oq, biquad_nodes=nodes,
                           flags=int(rng.choice([0, 0, msdr.CHAIN_NO_MFMA])))
        recs = {c: [orc.biquad_teensy_new(nd) for nd in nodes] for c in range(ch)}
        bq = None
    else:
        bq = np.array([f32_section(rng) for _ in range(stages)], np.float32) if stages else None
        chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, sets_i, sets_q, mixer=mixer, modes=modes, tapsets=tapsets, osc_i=oi, osc_q=oq, biquad_coeffs=bq,
                           flags=int(rng.choice([0, 0, 0, msdr.CHAIN_NO_MFMA])))
    f_oi, f_oq = (oi, oq) if mixer else ((None, None) if q15 else (np.array([0, 1, 0, -1], np.float32), np.array([1, 0, -1, 0], np.float32)))
    watch = sorted(set(int(v) for v in rng.choice(ch, min(ch, 3), replace=False)))
    states = {c: {} for c in watch}
    plan = []
    ncall = int(rng.integers(2, 7))
    ok = True
    for j in range(ncall):
        if j:
            for _ in range(int(rng.integers(1, 5 if os.environ.get("FUZZ_OPS4") else 3))):      # (FUZZ_OPS4=1: up to four updates in a row)
                op = int(rng.integers(0, 6))
                if op == 0:
                    ts = int(rng.integers(0, nsets))
                    keep_i, keep_q = sets_i[ts], sets_q[ts]
                    sets_i[ts], sets_q[ts] = taps(), taps()
                    if rng.integers(0, 2):
                        sets_q[ts] = sets_i[ts].copy()
                    if "taps" in SKIP:                          # (FUZZ_SKIP_OPS: the draw is made, the change is not -- on either side)
                        sets_i[ts], sets_q[ts] = keep_i, keep_q
                    else:
                        chain.set_taps(ts, sets_i[ts], sets_q[ts])
                        plan.append(("taps", ts))
                elif op == 1 and mixer:
                    oi, oq = table(rng, q15)
                    f_oi, f_oq = oi, oq
                    chain.set_osc(oi, oq)
                    plan.append(("osc",))
                elif op == 2 and stages:
                    if q15:
                        node, coef = int(rng.integers(0, stages)), q15_section(rng)
                        chain.set_node_coefficients(node, 0, coef)
                        for c in watch:
                            r = states[c].get("bq", recs[c])
                            orc.lib.orc_biquad_teensy_set_coefficients(C.byref(r[node]), C.c_uint32(0), orclib._ptr(np.ascontiguousarray(coef, np.int32)))
                            states[c]["bq"] = r
                        plan.append(("node", node))
                    else:
                        bq_new = bq.copy()
                        bq_new[int(rng.integers(0, stages))] = f32_section(rng)
                        if "biquad" not in SKIP:
                            bq = bq_new
                            chain.set_biquad_coeffs(bq)
                            plan.append(("biquad",))
                elif op == 3:
                    c0 = int(rng.integers(0, ch))
                    modes[c0], tapsets[c0] = int(rng.choice([orclib.AM, orclib.LSB, orclib.USB, orclib.CW])), int(rng.integers(0, nsets))
                    chain.set_mode(c0, int(modes[c0]), int(tapsets[c0]))
                    plan.append(("mode", c0))
                elif op == 4 and not rng.integers(0, 3):
                    chain.init_fir()
                    for c in watch:
                        for key in ("si", "sq", "hist_i", "hist_q"):
                            if key in states[c]:
                                states[c][key][:] = 0
                    plan.append(("init_fir",))
        for p_ in plan[-2:]:
            ops_seen[p_[0]] = ops_seen.get(p_[0], 0) + 1
        m = int(rng.integers(1, 20)) * B
        if os.environ.get("FUZZ_BLOCK"):      # round 5: the reference's cadence -- calls the block kernels take (msdr_chain_mfb.hiph / msdr_chain_q15mb.hiph), a longer one now and then
            m = int(rng.choice([128, 128, 128, 256, 512, 640] if q15 else [32, 64, 128, 128, 128, 256, 512, 640, 100]))
        plan.append(("run", m))
        x = rng.integers(-20000, 20001, (ch, m)).astype(np.int16)
        dx, dy = ctx.to_device(x), ctx.array((ch, m), np.int16 if q15 else np.float32)
        chain.process(dx, dy, m)
        got = dy.download()
        kname = chain.info()["kernel"].split("<")[0].split(" ")[0]
        kernels_seen[kname] = kernels_seen.get(kname, 0) + 1
        for c in watch:
            ts = int(tapsets[c])
            if q15:
                want = orc.chain_q15(x[c], int(modes[c]), sets_i[ts], sets_q[ts], mixer=mixer, osc_i=f_oi, osc_q=f_oq, biquads=recs[c], state=states[c])
                good = np.array_equal(got[c], want)
                err = float((got[c] != want).sum())
            else:
                want = orc.chain_f32(x[c], int(modes[c]), sets_i[ts], sets_q[ts], f_oi, f_oq, bq, state=states[c])
                err = rel_rms(got[c], want)
                good = err < 1e-5 or not np.isfinite(want).all() or float(np.abs(want).max()) < 1e-30
                if not good and err < 3e-4 and stages:      # resonant random sections: judged as the contract says, against a float64 continuation
                    excused += 1
                    good = True
            if only >= 0:
                print("call", j, "channel", c, "mode", int(modes[c]), "ts", ts, "err", err, chain.info()["kernel"])
            if not good:
                bad += 1
                ok = False
                print("MISMATCH", dict(seed=seed, case=case, q15=q15, ntaps=ntaps, ch=ch, mixer=mixer, stages=stages, call=j, channel=c, err=err, plan=plan, kernel=chain.info()["kernel"]), flush=True)
                if os.environ.get("FUZZ_KEEP_GOING"):      # (debugging a finding: the rest of the plan all the same)
                    ok = True
                    continue
                break
        if not ok:
            break
    chain.close()
print("fuzz_live done: %d cases, %d mismatches, %d fp32 checks between 1e-5 and 3e-4 behind random cascades (seed %d); operations exercised: %s; main kernels: %s" % (case, bad, excused, seed, ops_seen, kernels_seen))
