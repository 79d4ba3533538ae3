"""Q15 chain, msdr_chain_set_mode between calls, bit-exact against the oracle with a carried state (run from the repository root on
a GPU box):   gpurun -- python tests/debug/fuzz_retune_q15.py [seconds] [seed] [only_case]
Random tap counts / tap sets / mixers / biquad nodes / channel counts; several channels are retuned between calls (mode and
tap set), the FIR history and the biquad nodes' state carry over.  Each case draws from default_rng([seed, case])."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr  # noqa: E402  (imports torch first)
import orclib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
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
    ntaps = int(rng.integers(1, 130)) * 2
    nsets = int(rng.integers(1, 4))
    ch = int(rng.choice([1, 3, 64, 70]))
    amp = int(rng.choice([30, 3000, 32639]))
    ci = [rng.integers(-amp, amp + 1, ntaps).astype(np.int16) for _ in range(nsets)]
    cq = [rng.integers(-amp, amp + 1, ntaps).astype(np.int16) for _ in range(nsets)]
    modes = rng.integers(1, 5, ch).astype(np.int32)                      # AM, LSB, USB, CW (SYNCAM without the PLL flag = AM)
    tapsets = rng.integers(0, nsets, ch).astype(np.int32)
    mixer = int(rng.integers(0, 2))
    oi = oq = None
    if mixer:
        k = np.arange(B)
        oi = np.round(32767 * np.sin(2 * np.pi * k / 4)).astype(np.int16)
        oq = np.round(32767 * np.cos(2 * np.pi * k / 4)).astype(np.int16)
    nn = int(rng.integers(0, 3))
    corr = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
    specs = [[orc.biquad_design(orclib.BQ_LOWPASS, np.float32(5400 * corr), 0.54)], [orc.biquad_design(orclib.BQ_NOTCH, np.float32(3000 * corr), 15.0)]][:nn]
    sk = int(rng.integers(0, 2))
    ncall = int(rng.integers(2, 6))
    lens = [int(rng.integers(1, 12)) * B for _ in range(ncall)]
    x = rng.integers(-32768, 32768, (ch, sum(lens))).astype(np.int16) if rng.integers(0, 2) else rng.integers(-2000, 2001, (ch, sum(lens))).astype(np.int16)
    chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, ci, cq, mixer=mixer, modes=modes, tapsets=tapsets, osc_i=oi, osc_q=oq,
                       biquad_nodes=specs, sqrt_kind=msdr.SQRT_Q31 if sk else msdr.SQRT_F32)
    watch = sorted(set(int(c) for c in rng.choice(ch, min(ch, 4), replace=False)))
    states = {c: {} for c in watch}
    o, ok = 0, True
    for j, m in enumerate(lens):
        if j:
            for c in watch:
                if rng.integers(0, 2):
                    modes[c], tapsets[c] = int(rng.integers(1, 5)), int(rng.integers(0, nsets))
                    chain.set_mode(c, int(modes[c]), int(tapsets[c]))
        seg = np.ascontiguousarray(x[:, o:o + m])
        dx, dy = ctx.to_device(seg), ctx.array((ch, m), np.int16)
        chain.process(dx, dy, m)
        got = dy.download()
        for c in watch:
            want = orc.chain_q15(seg[c], int(modes[c]), ci[tapsets[c]], cq[tapsets[c]], mixer=mixer, osc_i=oi, osc_q=oq,
                                 sqrt_kind=orclib.SQRT_Q31 if sk else orclib.SQRT_F32,
                                 biquads=[orc.biquad_teensy_new(s) for s in specs], state=states[c])
            if not np.array_equal(got[c], want):
                bad += 1
                ok = False
                print("MISMATCH", dict(seed=seed, case=case, ntaps=ntaps, nsets=nsets, ch=ch, call=j, channel=c, mode=int(modes[c]), ts=int(tapsets[c]), mixer=mixer, nodes=nn,
                                       first=int(np.argmax(got[c] != want)), kernel=chain.info()["kernel"]), flush=True)
                break
        if not ok:
            break
        o += m
    chain.close()
print("fuzz_retune_q15 done: %d cases, %d mismatches (seed %d)" % (case, bad, seed))
