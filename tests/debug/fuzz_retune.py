"""fp32 chain, msdr_chain_set_mode between calls, every case reproducible on its own (run from the repository root on a GPU box):
    gpurun -- python tests/debug/fuzz_retune.py [seconds] [seed] [max_taps] [only_case]
Each case draws from default_rng([seed, case]); `only_case` replays one case verbosely."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr, rel_rms  # noqa: E402  (imports torch first)
import orclib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_taps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
only = int(sys.argv[4]) if len(sys.argv) > 4 else -1
orc = orclib.Oracle()
ctx = msdr.Context(0)
B = 128
t_end = time.time() + budget
case = bad = 0
while time.time() < t_end:
    case += 1
    if only >= 0:
        if case > 1:
            break
        case = only
    rng = np.random.default_rng([seed, case])
    ntaps = int(rng.integers(2, max_taps + 1))
    sets_i = [(rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32) for _ in range(2)]
    sets_q = [sets_i[0].copy(), (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)]
    stages = int(rng.integers(0, 3))
    corr = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
    bq = None
    if stages:
        rows = []
        for kind, f, q in ((orclib.BQ_LOWPASS, 5400 * corr, 0.54), (orclib.BQ_NOTCH, 3000 * corr, 15.0))[:stages]:
            c_ = orc.biquad_design(kind, np.float32(f), q).astype(np.float64) / 2 ** 30
            rows.append([c_[0], c_[1], c_[2], -c_[3], -c_[4]])
        bq = np.array(rows, np.float32)
    k = np.arange(B)
    mixer = int(rng.integers(0, 2))
    if mixer:
        oi = (np.round(32767 * np.sin(2 * np.pi * k / 4)).astype(np.int16) / 32768.0).astype(np.float32)
        oq = (np.round(32767 * np.cos(2 * np.pi * k / 4)).astype(np.int16) / 32768.0).astype(np.float32)
    else:
        oi, oq = np.array([0, 1, 0, -1], np.float32)[k % 4], np.array([1, 0, -1, 0], np.float32)[k % 4]
    ncall = int(rng.integers(2, 7))
    lens = [int(rng.integers(1, 30)) * B for _ in range(ncall)]
    x = rng.integers(-20000, 20001, (2, sum(lens))).astype(np.int16)
    mode0, ts0 = int(rng.choice([orclib.AM, orclib.LSB, orclib.USB])), int(rng.integers(0, 2))
    chain = msdr.Chain(ctx, msdr.ARITH_F32, 2, sets_i, sets_q, mixer=mixer, modes=np.array([mode0, orclib.LSB], np.int32), tapsets=np.array([ts0, 1], np.int32),
                       osc_i=oi if mixer else None, osc_q=oq if mixer else None, biquad_coeffs=bq)
    st0, st1, o = {}, {}, 0
    mode, ts = mode0, ts0
    plan = [(mode0, ts0, lens[0])]
    for j, m in enumerate(lens):
        if j and rng.integers(0, 3):
            mode, ts = int(rng.choice([orclib.AM, orclib.LSB, orclib.USB, orclib.CW])), int(rng.integers(0, 2))
            chain.set_mode(0, mode, ts)
        if j:
            plan.append((mode, ts, m))
        seg = np.ascontiguousarray(x[:, o:o + m])
        dx, dy = ctx.to_device(seg), ctx.array((2, m), np.float32)
        chain.process(dx, dy, m)
        got = dy.download()
        w0 = orc.chain_f32(seg[0], mode, sets_i[ts], sets_q[ts], oi, oq, bq, state=st0)
        w1 = orc.chain_f32(seg[1], orclib.LSB, sets_i[1], sets_q[1], oi, oq, bq, state=st1)
        e0, e1 = rel_rms(got[0], w0), rel_rms(got[1], w1)
        if only >= 0:
            d = np.abs(got[0].astype(np.float64) - w0)
            print("call", j, "mode", mode, "ts", ts, "len", m, "e0 %.3g e1 %.3g" % (e0, e1), "first bad sample", int(np.argmax(d > 1e-4 * max(np.abs(w0).max(), 1e-30))),
                  "max |d|", d.max(), "at", int(d.argmax()), "rms want", float(np.sqrt((w0.astype(np.float64) ** 2).mean())))
            if e0 > 1e-5:
                print("  got ", got[0][:8], "\n  want", w0[:8])
        if not (e0 < 1e-5 and e1 < 1e-5):
            bad += 1
            print("MISMATCH", dict(seed=seed, case=case, ntaps=ntaps, call=j, stages=stages, mixer=mixer, e0=e0, e1=e1, plan=plan, kernel=chain.info()["kernel"]))
            break
        o += m
    chain.close()
print("fuzz_retune done: %d cases, %d mismatches (seed %d, max_taps %d)" % (case, bad, seed, max_taps))
