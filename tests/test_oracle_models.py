"""Cross-checks for the oracle rows that CANNOT be pinned by running the reference here
("parity unpinned" in DESIGN.md): the Teensy integer biquad (A7), the freq_conv node (A2) and
the fp32 functions (A6, A8, fp32 chain).  Each is checked against an independent second model
written in this file (pure Python ints / numpy float64), and against the known-answer values
SURVEY.md's appendix recorded from its own probe build."""
import numpy as np
import pytest

import orclib

B = orclib.BLOCK
CORR = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0          # Minimal-SDR.ino:86


def s16(v):
    v &= 0xFFFF
    return v - 0x10000 if v & 0x8000 else v


def s32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def smlaw(acc, a, h):          # dspinst.h:235-249: acc + ((a * s16) >> 16), 32-bit wrap
    return s32(acc + ((a * h) >> 16))


def ssat_rshift(v, bits, sh):  # dspinst.h:34-51
    v >>= sh
    lim = 1 << (bits - 1)
    return max(-lim, min(lim - 1, v))


def py_biquad_packed(defn, data):
    """filter_biquad.cpp:33-82 followed literally in its two-samples-per-iteration packed form,
    in Python big ints (the C oracle uses the per-sample form: the two must agree)."""
    defn = [s32(int(v)) for v in defn]
    data = [int(v) for v in data]
    st = 0
    while True:
        b0, b1, b2, a1, a2 = defn[st:st + 5]
        bprev, aprev = defn[st + 5] & 0xFFFFFFFF, defn[st + 6] & 0xFFFFFFFF
        acc = defn[st + 7] & 0x3FFF
        for i in range(0, len(data), 2):
            in_lo, in_hi = data[i], data[i + 1]
            acc = smlaw(acc, b0, in_lo)
            acc = smlaw(acc, b1, s16(bprev >> 16))
            acc = smlaw(acc, b2, s16(bprev))
            acc = smlaw(acc, a1, s16(aprev >> 16))
            acc = smlaw(acc, a2, s16(aprev))
            out_lo = ssat_rshift(acc, 16, 14)
            acc &= 0x3FFF
            acc = smlaw(acc, b0, in_hi)
            acc = smlaw(acc, b1, in_lo)
            acc = smlaw(acc, b2, s16(bprev >> 16))
            acc = smlaw(acc, a1, s16(out_lo))
            acc = smlaw(acc, a2, s16(aprev >> 16))
            out_hi = ssat_rshift(acc, 16, 14)
            aprev = ((out_hi & 0xFFFF) << 16) | (out_lo & 0xFFFF)
            acc &= 0x3FFF
            bprev = ((in_hi & 0xFFFF) << 16) | (in_lo & 0xFFFF)
            data[i], data[i + 1] = out_lo, out_hi
        flag = defn[st + 7] & 0x80000000
        defn[st + 7] = s32(acc | flag)
        defn[st + 6], defn[st + 5] = s32(aprev), s32(bprev)
        st += 8
        if not flag:
            break
    return defn, np.array(data, np.int16)


# ---------------------------------------------------------------- A7 ----------------------
def test_biquad_designer_known_answers(orc):
    # SURVEY.md appendix (probe values): stage words after setCoefficients (a1,a2 negated)
    lp = orc.biquad_design(orclib.BQ_LOWPASS, np.float32(5400 * CORR), 0.54)
    b = orc.biquad_teensy_new([lp])
    assert list(b.definition[0:5]) == [236552419, 473104839, 236552419, 175469220, -47937074]
    nt = orc.biquad_design(orclib.BQ_NOTCH, np.float32(3000 * CORR), 15.0)
    b2 = orc.biquad_teensy_new([lp, nt])
    assert list(b2.definition[8:13]) == [1049016272, -1483533003, 1049016272, 1483533003, -1024290721]
    assert (b2.definition[7] & 0xFFFFFFFF) == 0x80000000      # "another stage follows"
    assert (b2.definition[15] & 0xFFFFFFFF) == 0


def test_biquad_set_coefficients_ignores_stage_4(orc):
    b = orc.biquad_teensy_new([[1, 2, 3, 4, 5]])
    before = list(b.definition)
    orc.lib.orc_biquad_teensy_set_coefficients(orclib.C.byref(b), 4,
                                               np.array([9, 9, 9, 9, 9], np.int32).ctypes.data_as(orclib._p))
    assert list(b.definition) == before


def _nodes(orc, n_stage):
    qs = [0.54, 1.3, 0.54, 1.3][:n_stage]                     # .ino:393-399 Linkwitz-Riley set
    return [orc.biquad_design(orclib.BQ_LOWPASS, np.float32(6000 * 0.9 * CORR), q) for q in qs]


@pytest.mark.parametrize("n_stage", [1, 2, 4])
@pytest.mark.parametrize("amp", [3000, 32767])
def test_biquad_update_matches_packed_python_model(orc, n_stage, amp):
    rng = np.random.default_rng(n_stage * 7 + amp)
    b = orc.biquad_teensy_new(_nodes(orc, n_stage))
    defn = list(b.definition)
    for blk in range(6):
        x = rng.integers(-amp, amp + 1, B).astype(np.int16)
        if amp == 32767 and blk == 2:
            x[:] = 32767                                      # drive the output into saturation
        y = orc.biquad_teensy_update(b, x)
        defn, want = py_biquad_packed(defn, x)
        assert np.array_equal(y, want)
        assert [s32(v) for v in b.definition] == defn


def test_biquad_notch_rejects_fs8(orc):
    # .ino:356: setNotch(0, fs/8*CORR, 15): a tone at fs/8 must be strongly attenuated
    nt = orc.biquad_design(orclib.BQ_NOTCH, np.float32(3000 * CORR), 15.0)
    b = orc.biquad_teensy_new([nt])
    n = np.arange(40 * B)
    x = np.round(10000 * np.sin(2 * np.pi * n / 8)).astype(np.int16)
    y = np.concatenate([orc.biquad_teensy_update(b, x[o:o + B]) for o in range(0, x.size, B)])
    assert np.abs(y[-4 * B:].astype(int)).max() < 200


# ---------------------------------------------------------------- A2 ----------------------
def _sat(v):
    return np.clip(v, -32768, 32767)


@pytest.mark.parametrize("direction", [0, 1])
@pytest.mark.parametrize("passthrough", [0, 1])
def test_freqconv_q15_vs_numpy(orc, direction, passthrough):
    rng = np.random.default_rng(direction * 2 + passthrough)
    i = rng.integers(-32768, 32768, B).astype(np.int16)
    q = rng.integers(-32768, 32768, B).astype(np.int16)
    n = np.arange(B)
    oi = np.round(32767 * np.sin(2 * np.pi * 5 * n / B)).astype(np.int16)
    oq = np.round(32767 * np.cos(2 * np.pi * 5 * n / B)).astype(np.int16)
    oi[3], oq[3], i[3], q[3] = -32768, -32768, -32768, -32768  # (-1)*(-1) saturates
    gi, gq = orc.freqconv_q15(i, q, oi, oq, direction, passthrough)
    if not passthrough:                                        # freq_conv.cpp:49-56 (inverted flag)
        assert np.array_equal(gi, i) and np.array_equal(gq, q)
        return
    I, Q, OI, OQ = (v.astype(np.int64) for v in (i, q, oi, oq))
    m = lambda a, b: _sat((a * b) >> 15)
    if direction == 0:
        wi, wq = _sat(m(I, OQ) + m(Q, OI)), _sat(m(Q, OQ) - m(I, OI))
    else:
        wq, wi = _sat(m(Q, OQ) + m(I, OI)), _sat(m(I, OQ) - m(Q, OI))
    assert np.array_equal(gi, wi) and np.array_equal(gq, wq)


def test_freqconv_reduces_to_fs4_mixer(orc):
    """with osc tables = Fs/4 patterns the node equals the inline mixer A1 up to the
    32767/32768 scaling of +1."""
    x = np.random.default_rng(2).integers(-20000, 20000, B).astype(np.int16)
    cos4 = np.tile(np.array([32767, 0, -32768, 0], np.int16), B // 4)
    sin4 = np.tile(np.array([0, 32767, 0, -32768], np.int16), B // 4)
    gi, gq = orc.freqconv_q15(x, np.zeros(B, np.int16), sin4, cos4, 1, 1)
    mi, mq = orc.mix_fs4(x)
    assert np.abs(gi.astype(int) - mi).max() <= 1 and np.abs(gq.astype(int) - mq).max() <= 1


# ---------------------------------------------------------------- A6 / A8 fp32 ------------
def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-300))


@pytest.mark.parametrize("ntaps", [1, 2, 61, 100, 256, 512])
def test_fir_f32_vs_float64(orc, ntaps):
    rng = np.random.default_rng(ntaps)
    h = (rng.standard_normal(ntaps) / ntaps).astype(np.float32)
    x = rng.uniform(-1, 1, 5 * B + 13).astype(np.float32)
    y = orc.fir_f32_blocks(h, x, B)
    w = np.concatenate([np.zeros(ntaps - 1), x.astype(np.float64)])
    want = np.array([np.dot(w[n:n + ntaps], h.astype(np.float64)) for n in range(x.size)])
    assert rel_rms(y, want) < 2e-6


@pytest.mark.parametrize("stages", [1, 2, 4])
def test_biquad_df1_f32_vs_scipy(orc, stages):
    from scipy.signal import lfilter
    rng = np.random.default_rng(stages)
    qs = [0.54, 1.3, 0.54, 1.3][:stages]
    coeffs = []
    for q in qs:
        c = orc.biquad_design(orclib.BQ_LOWPASS, np.float32(5400 * CORR), q).astype(np.float64) / 2 ** 30
        coeffs.append([c[0], c[1], c[2], -c[3], -c[4]])        # CMSIS convention: feedback ADDED
    x = rng.uniform(-1, 1, 6 * B).astype(np.float32)
    y = orc.biquad_df1_blocks(np.array(coeffs, np.float32), x, B)
    want = x.astype(np.float64)
    for c in np.array(coeffs, np.float32).astype(np.float64):
        want = lfilter(c[:3], [1.0, -c[3], -c[4]], want)
    assert rel_rms(y, want) < 1e-5


def test_chain_f32_tracks_q15_chain(orc, golden):
    """SURVEY 8c cross-check (1): the fp32 chain fed the same int16 IF and the Q15-scaled taps must
    agree with the bit-exact q15 chain to within the q15 truncation bound (a few LSB)."""
    cos4 = np.array([1, 0, -1, 0], np.float32)
    sin4 = np.array([0, 1, 0, -1], np.float32)
    for sn in ("am", "tones", "noise"):
        x = golden["chain/x_" + sn]
        for mn, mode, ti, tq in (("AM", orclib.AM, "fir/taps_am102", "fir/taps_am102"),
                                 ("LSB", orclib.LSB, "taps/FIR_SSB_I_coeffs", "taps/FIR_SSB_Q_coeffs"),
                                 ("USB", orclib.USB, "taps/FIR_SSB_I_coeffs", "taps/FIR_SSB_Q_coeffs")):
            hi = golden[ti].astype(np.float32) / 32768
            hq = golden[tq].astype(np.float32) / 32768
            y = orc.chain_f32(x, mode, hi, hq, sin4, cos4, None, in_scale=1.0)
            want = golden["chain/%s_%s_audio" % (sn, mn)].astype(np.float64)
            assert np.abs(y - want).max() <= 3.0, (sn, mn)


def test_chain_f32_state_carry_equals_one_shot(orc, golden):
    x = golden["chain/x_tones"]
    n = np.arange(B)
    oi = np.sin(2 * np.pi * 32 * n / B).astype(np.float32)
    oq = np.cos(2 * np.pi * 32 * n / B).astype(np.float32)
    hi = golden["taps/FIR_SSB_I_coeffs"].astype(np.float32) / 32768
    hq = golden["taps/FIR_SSB_Q_coeffs"].astype(np.float32) / 32768
    bq = np.array([[0.2, 0.4, 0.2, 0.3, -0.1], [0.9, -1.2, 0.9, 1.2, -0.85]], np.float32)
    one = orc.chain_f32(x, orclib.LSB, hi, hq, oi, oq, bq)
    st = {}
    parts = [orc.chain_f32(x[o:o + 300], orclib.LSB, hi, hq, oi, oq, bq, state=st) for o in range(0, x.size, 300)]
    assert np.array_equal(np.concatenate(parts), one)


def test_batch_drivers_equal_single_channel(orc, golden):
    x = np.stack([golden["chain/x_am"], golden["chain/x_tones"], golden["chain/x_noise"]])
    hi, hq = golden["taps/FIR_SSB_I_coeffs"], golden["taps/FIR_SSB_Q_coeffs"]
    lp = orc.biquad_teensy_new([orc.biquad_design(orclib.BQ_LOWPASS, np.float32(5400 * CORR), 0.54)])
    modes = np.array([orclib.LSB, orclib.USB, orclib.AM], np.int32)
    out, used = orc.chain_q15_batch(x, modes, hi, hq, biquads=[lp], threads=2)
    assert used == 2
    for c in range(3):
        assert np.array_equal(out[c], orc.chain_q15(x[c], modes[c], hi, hq, biquads=[lp]))
    cos4 = np.array([1, 0, -1, 0], np.float32)
    sin4 = np.array([0, 1, 0, -1], np.float32)
    hif, hqf = hi.astype(np.float32) / 32768, hq.astype(np.float32) / 32768
    bq = np.array([[0.2, 0.4, 0.2, 0.3, -0.1]], np.float32)
    outf, _ = orc.chain_f32_batch(x, modes, hif, hqf, sin4, cos4, bq, threads=2)
    for c in range(3):
        assert np.array_equal(outf[c], orc.chain_f32(x[c], modes[c], hif, hqf, sin4, cos4, bq))
