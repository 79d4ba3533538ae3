"""Q15 chain with the SYNCAM PLL (MSDR_CHAIN_SYNCAM_PLL) and the LMS notch / noise reduction (msdr_chain_set_anr), random tap
counts / channel counts / modes / mixers / call lengths, bit-exact against the composed oracle (FIR pair -> PLL or demodulator
-> LMS filter -> biquad node):   gpurun -- python tests/debug/fuzz_pll_anr.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr  # noqa: E402  (imports torch first)
import orclib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc = orclib.Oracle()
ctx = msdr.Context(0)
B = 128
CORR = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
t_end = time.time() + budget
case = bad = 0
while time.time() < t_end:
    case += 1
    rng = np.random.default_rng([seed, case])
    ntaps = int(rng.integers(1, 80)) * 2
    ch = int(rng.choice([1, 3, 8, 64, 70]))
    nblk = int(rng.integers(2, 30))
    n = nblk * B
    t = np.arange(n)
    if rng.integers(0, 2):
        taps_i = msdr.calc_fir_coeffs(ntaps, float(rng.uniform(1500, 4000)))[:ntaps]
        taps_q = taps_i.copy()
    else:
        amp = int(rng.choice([300, 3000]))
        taps_i, taps_q = rng.integers(-amp, amp + 1, ntaps).astype(np.int16), rng.integers(-amp, amp + 1, ntaps).astype(np.int16)
    x = np.empty((ch, n), np.int16)
    for c in range(ch):
        car = 6000 + rng.uniform(-80, 80)
        x[c] = (rng.uniform(500, 14000) * (1 + 0.5 * np.sin(2 * np.pi * rng.uniform(100, 900) * t / 24000)) * np.cos(2 * np.pi * car * t / 24000 + c)
                + rng.uniform(0, 2000) * np.cos(2 * np.pi * 7000 * t / 24000) + rng.integers(-200, 201, n)).clip(-32768, 32767).astype(np.int16)
    modes = rng.integers(0, 5, ch).astype(np.int32)
    anr_on = rng.integers(0, 3, ch).astype(np.int32)
    use_anr = bool(rng.integers(0, 2))
    nodes = [[msdr.biquad_design(msdr.BQ_LOWPASS, np.float32(6000 * 0.9 * CORR), 0.54)]] if rng.integers(0, 2) else []
    chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, taps_i, taps_q, modes=modes, biquad_nodes=nodes, flags=msdr.CHAIN_SYNCAM_PLL)
    if use_anr:
        chain.set_anr(anr_on)
    got = np.empty((ch, n), np.int16)
    b0 = 0
    while b0 < nblk:
        m = int(min(nblk - b0, rng.choice([1, 2, 5, 9, nblk]))) * B
        dx, dy = ctx.to_device(np.ascontiguousarray(x[:, b0 * B:b0 * B + m])), ctx.array((ch, m), np.int16)
        chain.process(dx, dy, m)
        got[:, b0 * B:b0 * B + m] = dy.download()
        b0 += m // B
    for c in sorted(set(int(v) for v in rng.choice(ch, min(ch, 5), replace=False))):
        if modes[c] == orclib.SYNCAM:
            _, i_f, q_f = orc.chain_q15(x[c], orclib.AM, taps_i, taps_q, want_iq=True)
            audio = orc.syncam_q15(orc.syncam_new(), i_f, q_f)
        else:
            audio = orc.chain_q15(x[c], int(modes[c]), taps_i, taps_q)
        if use_anr:
            audio = orc.anr_q15(orc.anr_new(), int(anr_on[c]), audio)
        want = orc.biquad_teensy_update(orc.biquad_teensy_new(nodes[0]), audio) if nodes else audio
        if not np.array_equal(got[c], want):
            bad += 1
            print("MISMATCH", dict(seed=seed, case=case, ntaps=ntaps, ch=ch, nblk=nblk, channel=c, mode=int(modes[c]), anr=int(anr_on[c]) if use_anr else -1, nodes=len(nodes),
                                   first=int(np.argmax(got[c] != want)), kernel=chain.info()["kernel"]), flush=True)
            break
    chain.close()
print("fuzz_pll_anr done: %d cases, %d mismatches (seed %d)" % (case, bad, seed))
