"""Randomised differential test of the matrix-core kernels against the oracle (run from the repository root on a GPU box):
    gpurun -- python tests/debug/fuzz_kernels.py [seconds] [seed]
Q15 chain (random tap counts, tap sets, modes, mixers, call lengths, biquad nodes), arm_fir_fast_q15 stage, arm_fir_f32 stage, spectrum FFT."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr, rel_rms  # noqa: E402  (imports torch first)
import orclib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = orclib.Oracle()
ctx = msdr.Context(0)
B = 128
t_end = time.time() + budget
cases = bad = inherent = 0
seen = {}


def truth64(x, mode, hi, hq, oi, oq, bq):
    """orc_chain_f32 (oracle/msdr_oracle.c) with every operation in float64."""
    from scipy.signal import lfilter
    n = np.arange(x.size)
    xf = x.astype(np.float64) * (1.0 / 32768)
    wi, wq = xf * oq.astype(np.float64)[n % oq.size], xf * oi.astype(np.float64)[n % oi.size]
    ai = lfilter(hi.astype(np.float64)[::-1], [1.0], wi)
    aq = lfilter(hq.astype(np.float64)[::-1], [1.0], wq)
    d = ai - aq if mode == orclib.LSB else ai + aq if mode == orclib.USB else np.sqrt(ai * ai + aq * aq)
    if bq is not None:
        for c in np.asarray(bq, np.float64):
            d = lfilter(c[:3], [1.0, -c[3], -c[4]], d)
    return d


def rand_x(ch, n):
    kind = rng.integers(0, 4)
    if kind == 0:
        return rng.integers(-32768, 32768, (ch, n)).astype(np.int16)
    if kind == 1:
        return rng.integers(-300, 301, (ch, n)).astype(np.int16)
    if kind == 2:
        x = rng.integers(-20000, 20001, (ch, n)).astype(np.int16)
        x[:, ::int(rng.integers(2, 9))] = -32768
        return x
    return (np.sign(rng.standard_normal((ch, n))) * 32767).astype(np.int16)


def splits(n):
    out, o = [], 0
    while o < n:
        m = int(min(n - o, rng.choice([1, 2, 3, 7, 128, 130, 1000, 1024, 1025, 4096, 5000, n])))
        out.append((o, m)); o += m
    return out


while time.time() < t_end:
    which = rng.integers(0, 6)
    cases += 1
    if which == 0:      # Q15 chain
        ntaps = int(rng.integers(1, 257)) * 2
        nsets = int(rng.integers(1, 4))
        ch = int(rng.choice([1, 3, 64, 70, 128]))
        n = int(rng.integers(2, 60)) * B
        amp = int(rng.choice([30, 3000, 32639]))
        ci = [rng.integers(-amp, amp + 1, ntaps).astype(np.int16) for _ in range(nsets)]
        cq = [rng.integers(-amp, amp + 1, ntaps).astype(np.int16) for _ in range(nsets)]
        modes = rng.integers(0, 5, ch).astype(np.int32)
        tapsets = rng.integers(0, nsets, ch).astype(np.int32)
        mixer = int(rng.integers(0, 2))
        oi = oq = None
        if mixer:
            kind = int(rng.integers(0, 4))
            if kind < 2:            # the fs/4 pattern (zeros at one parity): the zero-skipping layout
                a, b, c_, d = [int(v) for v in rng.integers(-32768, 32768, 4)]
                q4, i4 = ([a, 0, b, 0], [0, c_, 0, d]) if kind == 0 else ([0, a, 0, b], [c_, 0, d, 0])
                oq, oi = np.array(q4, np.int16)[np.arange(B) % 4], np.array(i4, np.int16)[np.arange(B) % 4]
            elif kind == 2:         # a sampled oscillator of period 8 .. 128 (full-rate layout), q15-rounded, now and then with full-scale entries
                P_ = int(rng.choice([8, 16, 32, 64, 128]))
                k_ = np.arange(B)
                amp_ = 32767 if rng.integers(0, 2) else int(rng.integers(1000, 32768))
                oi = np.round(amp_ * np.sin(2 * np.pi * k_ / P_)).astype(np.int16)
                oq = np.round(amp_ * np.cos(2 * np.pi * k_ / P_)).astype(np.int16)
                if rng.integers(0, 3) == 0:
                    oi[rng.integers(0, B, 3)] = -32768; oq[rng.integers(0, B, 3)] = -32768
            else:                   # an unstructured 128-entry table
                oi, oq = rng.integers(-32768, 32768, B).astype(np.int16), rng.integers(-32768, 32768, B).astype(np.int16)
            if ntaps >= 248:
                ntaps = int(rng.integers(1, 124)) * 2      # (the general-table layouts take < 248 taps; longer ones run the vector-ALU kernel: slow to fuzz)
                ci = [rng.integers(-amp, amp + 1, ntaps).astype(np.int16) for _ in range(nsets)]
                cq = [rng.integers(-amp, amp + 1, ntaps).astype(np.int16) for _ in range(nsets)]
        nodes = []
        even_only = False
        if rng.integers(0, 2):
            lp = orc.biquad_design(orclib.BQ_LOWPASS, np.float32(rng.uniform(500, 9000)), float(rng.uniform(0.4, 3)))
            nt = orc.biquad_design(orclib.BQ_NOTCH, np.float32(rng.uniform(500, 9000)), float(rng.uniform(1, 20)))
            nodes = [[lp], [nt]][:int(rng.integers(1, 3))]
            even_only = True
        sk = int(rng.integers(0, 2))
        x = rand_x(ch, n)
        chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, ci, cq, mixer=mixer, modes=modes, tapsets=tapsets, osc_i=oi, osc_q=oq,
                           biquad_nodes=nodes, sqrt_kind=sk)
        got = np.empty_like(x)
        # ragged calls; AudioFilterBiquad processes sample pairs, so with nodes every call length is made even
        o = 0
        for _, m in splits(n):
            if o >= n:
                break
            m = min(m, n - o)
            if even_only and (m & 1):
                m = m + 1 if o + m + 1 <= n else m - 1
            if m <= 0:
                m = n - o
            dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.array((ch, m), np.int16)
            chain.process(dx, dy, m)
            got[:, o:o + m] = dy.download()
            o += m
        for c in rng.choice(ch, min(ch, 4), replace=False):
            want = orc.chain_q15(x[c], modes[c], ci[tapsets[c]], cq[tapsets[c]], mixer=mixer, osc_i=oi, osc_q=oq, sqrt_kind=sk,
                                 biquads=[orc.biquad_teensy_new(nd) for nd in nodes])
            if not np.array_equal(got[c], want):
                bad += 1
                print("MISMATCH q15 chain", dict(seed=seed, case=cases, ntaps=ntaps, ch=ch, n=n, mixer=mixer, mode=int(modes[c]), nodes=len(nodes), sk=sk,
                                                  kernel=chain.info()["kernel"], first=int(np.nonzero(got[c] != want)[0][0])))
                break
        seen[chain.info()["kernel"]] = seen.get(chain.info()["kernel"], 0) + 1
        chain.close()
    elif which == 1:    # arm_fir_fast_q15 stage
        ntaps = int(rng.integers(1, 257)) * 2
        ch, n = int(rng.choice([1, 5, 70])), int(rng.integers(1, 80)) * B
        taps = rng.integers(-int(rng.choice([50, 5000, 32639])), int(rng.choice([50, 5000, 32639])) + 1, ntaps).astype(np.int16)
        x = rand_x(ch, n)
        fir = msdr.FirQ15(ctx, taps, ch)
        got = np.empty_like(x)
        for o, m in splits(n):
            dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.array((ch, m), np.int16)
            fir.process(dx, dy, m)
            got[:, o:o + m] = dy.download()
        c = int(rng.integers(0, ch))
        rc, want = orc.fir_q15_blocks(taps, x[c], B)
        if not np.array_equal(got[c], want):
            bad += 1
            print("MISMATCH fir q15", dict(seed=seed, case=cases, ntaps=ntaps, ch=ch, n=n))
        fir.close()
    elif which == 3:    # fp32 chain (matrix-core kernels for period-1/2/4 oscillators, else the general kernel)
        ntaps = int(rng.integers(2, 300))
        ch = int(rng.choice([1, 3, 40]))
        n = int(rng.integers(2, 80)) * B
        hi = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
        hq = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
        modes = rng.choice([orclib.AM, orclib.LSB, orclib.USB, orclib.CW], ch).astype(np.int32)
        if (modes == orclib.AM).any() or (modes == orclib.CW).any():
            hq = hi.copy() if rng.integers(0, 4) else hq
        mixer = int(rng.integers(0, 2))
        P = int(rng.choice([1, 2, 4, 8, 16, 32, 64, 128, 0]))
        k = np.arange(B)
        if mixer and P == 0:          # an unstructured table (full-rate layout)
            oi, oq = rng.uniform(-1, 1, B).astype(np.float32), rng.uniform(-1, 1, B).astype(np.float32)
        elif mixer:
            oi = (np.round(32767 * np.sin(2 * np.pi * k / P)).astype(np.int16) / 32768.0).astype(np.float32)
            oq = (np.round(32767 * np.cos(2 * np.pi * k / P)).astype(np.int16) / 32768.0).astype(np.float32)
        else:
            oi, oq = np.array([0, 1, 0, -1], np.float32)[k % 4], np.array([1, 0, -1, 0], np.float32)[k % 4]
        stages = int(rng.integers(0, 5))
        bq = None
        if stages:
            rows = []
            for _ in range(stages):
                kind = int(rng.choice([orclib.BQ_LOWPASS, orclib.BQ_NOTCH, orclib.BQ_HIGHPASS]))
                c_ = orc.biquad_design(kind, np.float32(rng.uniform(800, 9000)), float(rng.uniform(0.5, 8))).astype(np.float64) / 2 ** 30
                rows.append([c_[0], c_[1], c_[2], -c_[3], -c_[4]])
            bq = np.array(rows, np.float32)
        x = rand_x(ch, n)
        # envelope channels behind the Fs/4 mixer with hI == hQ: every second case through chain_amtr_kernel whatever the tap count
        if rng.integers(0, 2):
            os.environ["MSDR_AMTR"] = "1"
        else:
            os.environ.pop("MSDR_AMTR", None)
        chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, hi, hq, mixer=mixer, modes=modes, osc_i=oi if mixer else None, osc_q=oq if mixer else None,
                           biquad_coeffs=bq, time_segments=int(rng.choice([0, 0, 1, 3])))
        got = np.empty((ch, n), np.float32)
        o = 0
        for _, m in splits(n):
            if o >= n:
                break
            m = min(m, n - o)
            dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.array((ch, m), np.float32)
            chain.process(dx, dy, m)
            got[:, o:o + m] = dy.download()
            o += m
        for c in rng.choice(ch, min(ch, 3), replace=False):
            want = orc.chain_f32(x[c], modes[c], hi, hq, oi, oq, bq)
            err = rel_rms(got[c], want)
            # a cascade that removes most of its input turns the 2e-7 agreement in front of it into a larger RELATIVE error of what is
            # left (any fp32 evaluation does, the oracle included): the bound is referred to the cascade's input level
            tol = 1e-5
            if bq is not None:
                pre = orc.chain_f32(x[c], modes[c], hi, hq, oi, oq, None)
                tol *= max(1.0, float(np.sqrt((pre.astype(np.float64) ** 2).mean() / max((want.astype(np.float64) ** 2).mean(), 1e-300))))
            if not err < tol:
                # beyond the tolerance of the fp32 oracle: a defect only if the library is further from a float64 evaluation than the
                # oracle itself is (resonant / deep-stop-band cascades: any fp32 evaluation is noisy; see fuzz_f32_truth.py)
                t64 = truth64(x[c], int(modes[c]), hi, hq, oi, oq, bq)
                if rel_rms(got[c], t64) <= 2 * rel_rms(want, t64) + (msdr.biquad_cascade_info(bq)[1] if bq is not None else 0.0) + 1e-6:
                    inherent += 1
                    continue
                bad += 1
                print("MISMATCH f32 chain", dict(seed=seed, case=cases, ntaps=ntaps, ch=ch, n=n, mixer=mixer, P=P, mode=int(modes[c]), stages=stages, err=err,
                                                  kernel=chain.info()["kernel"]))
                break
        seen[chain.info()["kernel"]] = seen.get(chain.info()["kernel"], 0) + 1
        chain.close()
    elif which == 4:    # fp32 chain, msdr_chain_set_mode between calls (FIR history and cascade state carry over; folded-numerator modes are entered / left)
        ntaps = int(rng.integers(2, 200))
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
        x = rand_x(2, sum(lens))
        mode0, ts0 = int(rng.choice([orclib.AM, orclib.LSB, orclib.USB])), int(rng.integers(0, 2))
        chain = msdr.Chain(ctx, msdr.ARITH_F32, 2, sets_i, sets_q, mixer=mixer, modes=np.array([mode0, orclib.LSB], np.int32), tapsets=np.array([ts0, 1], np.int32),
                           osc_i=oi if mixer else None, osc_q=oq if mixer else None, biquad_coeffs=bq)
        st0, st1, o, ok = {}, {}, 0, True
        mode, ts = mode0, ts0
        for j, m in enumerate(lens):
            if j and rng.integers(0, 3):
                mode, ts = int(rng.choice([orclib.AM, orclib.LSB, orclib.USB, orclib.CW])), int(rng.integers(0, 2))
                chain.set_mode(0, mode, ts)
            seg = np.ascontiguousarray(x[:, o:o + m])
            dx, dy = ctx.to_device(seg), ctx.array((2, m), np.float32)
            chain.process(dx, dy, m)
            got = dy.download()
            w0 = orc.chain_f32(seg[0], mode, sets_i[ts], sets_q[ts], oi, oq, bq, state=st0)
            w1 = orc.chain_f32(seg[1], orclib.LSB, sets_i[1], sets_q[1], oi, oq, bq, state=st1)
            e0, e1 = rel_rms(got[0], w0), rel_rms(got[1], w1)
            if not (e0 < 1e-5 and e1 < 1e-5):
                bad += 1
                print("MISMATCH f32 retune", dict(seed=seed, case=cases, ntaps=ntaps, call=j, mode=mode, ts=ts, stages=stages, mixer=mixer, e0=e0, e1=e1, kernel=chain.info()["kernel"]))
                break
            o += m
        chain.close()
    elif which == 5:    # spectrum FFT (arm_rfft_q15, 128 points) + column heights
        nfft = int(rng.choice([1, 5, 16, 33, 200]))
        stride = 128 + 8 * int(rng.integers(0, 3))
        kind = int(rng.integers(0, 4))
        k_ = np.arange(128)
        if kind == 0:
            xs = rng.integers(-32768, 32768, (nfft, 128))
        elif kind == 1:
            xs = rng.integers(-int(rng.choice([3, 300, 9000])), int(rng.choice([3, 300, 9000])) + 1, (nfft, 128))
        elif kind == 2:     # few strong lines: saturating adds
            xs = np.zeros((nfft, 128))
            for _ in range(int(rng.integers(1, 4))):
                xs += rng.uniform(8000, 45000) * np.cos(2 * np.pi * rng.integers(0, 64, (nfft, 1)) * k_ / 128 + rng.uniform(0, 6.3, (nfft, 1)))
            xs = np.clip(np.round(xs), -32768, 32767)
        else:               # two-level signals
            xs = np.where(rng.integers(0, 2, (nfft, 128)) == 1, 32767, -32768) * rng.integers(0, 2, (nfft, 128))
        xs = xs.astype(np.int16)
        buf = np.zeros((nfft, stride), np.int16)
        buf[:, :128] = xs
        o_, c_ = ctx.array((nfft, 256), np.int16), ctx.array((nfft, 128), np.uint8)
        msdr.rfft128_q15(ctx, ctx.to_device(buf), stride, nfft, o_, c_)
        go, gc = o_.download(), c_.download()
        for f in rng.choice(nfft, min(nfft, 6), replace=False):
            want, _ = orc.rfft128_q15(xs[f])
            if not (np.array_equal(go[f], want) and np.array_equal(gc[f, :127], orc.spectrum_columns(want)) and gc[f, 127] == 0):
                bad += 1
                print("MISMATCH spectrum", dict(seed=seed, case=cases, nfft=nfft, stride=stride, kind=kind, f=int(f)))
                break
    else:               # arm_fir_f32 stage
        ntaps = int(rng.integers(1, 513))
        ch, n = int(rng.choice([1, 5, 70])), int(rng.integers(1, 80)) * B
        h = (rng.standard_normal(ntaps) * 10.0 ** rng.uniform(-3, 1)).astype(np.float32)
        x = (rng.standard_normal((ch, n)) * 10.0 ** rng.uniform(-6, 6)).astype(np.float32)
        if rng.integers(0, 2):
            x[:, n // 2:] *= np.float32(10.0 ** rng.uniform(-4, 4))
        fir = msdr.FirF32(ctx, h, ch)
        got = np.empty_like(x)
        for o, m in splits(n):
            dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.array((ch, m), np.float32)
            fir.process(dx, dy, m)
            got[:, o:o + m] = dy.download()
        c = int(rng.integers(0, ch))
        want = np.convolve(x[c].astype(np.float64), h[::-1].astype(np.float64))[:n]     # pCoeffs are stored time-reversed
        err = rel_rms(got[c], want)
        if not err < 3e-6:
            bad += 1
            print("MISMATCH fir f32", dict(seed=seed, case=cases, ntaps=ntaps, ch=ch, n=n, err=err))
        fir.close()
    if cases % 50 == 0:
        print("cases", cases, "bad", bad, flush=True)
print("kernels exercised by the chain cases:", seen)
print("fuzz done: %d cases, %d mismatches (seed %d); %d fp32-chain checks beyond the oracle tolerance where the oracle is as far from float64" % (cases, bad, seed, inherent))
sys.exit(1 if bad else 0)
