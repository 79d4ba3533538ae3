"""Pin the CPU oracle (oracle/msdr_oracle.c) against the reference.

Two anchors:
  * `ref`    -- oracle/_ref/libmsdr_ref.so, the reference's own sources compiled in this container
               (skipped where /root/reference is absent, e.g. on the GPU box if not prebuilt);
  * `golden` -- tests/golden/golden.npz, outputs of that compiled reference, committed.
Bit-exact everywhere (integer path)."""
import numpy as np
import pytest

import goldenlib
import orclib

B = orclib.BLOCK


def test_golden_manifest_intact(golden):
    assert goldenlib.verify(golden) == []


# ---------------------------------------------------------------- A9 designer -------------
def test_designer_matches_golden(orc, golden):
    for (n, fc, a, t, dfc) in golden.meta["design_cases"]:
        for pid in (False, True):
            want = golden["design/n%d_fc%d_a%d_t%d_d%d_%s" % (n, fc, a, t, dfc, "pid" if pid else "pif")]
            got = orc.calc_fir_coeffs(n, fc, a, t, dfc, 24000.0, pi_double=pid, room=want.size)
            assert np.array_equal(got, want), (n, fc, a, t, dfc, pid)


def test_designer_known_answers_from_survey(orc):
    # calc_FIR_coeffs(...,102, 2800, 70, 0, 0, 24000)
    c = orc.calc_fir_coeffs(102, 2800)[:102].astype(int)
    assert list(c[49:54]) == [5161, 6970, 7645, 6970, 5161]
    assert c[0] == 0 and list(c[1:4]) == [-2, -3, -2]
    assert c.sum() == 32761 and np.abs(c).sum() == 59529


def test_designer_matches_reference_live(orc, ref):
    rng = np.random.default_rng(11)
    for _ in range(60):
        n = int(rng.integers(4, 300)) & ~1
        fc = float(rng.integers(100, 9000))
        a = float(rng.choice([10.0, 30.0, 45.5, 50.0, 70.0, 90.0]))
        t = int(rng.integers(0, 4))
        dfc = float(rng.integers(50, 2000))
        for pid in (False, True):
            want = ref.calc_fir_coeffs(n, fc, a, t, dfc, 24000.0, pi_double=pid)
            got = orc.calc_fir_coeffs(n, fc, a, t, dfc, 24000.0, pi_double=pid)
            assert np.array_equal(got, want), (n, fc, a, t, dfc, pid)
    for n in (8, 32, 64):  # Hilbert branch (type 4), never used by the app
        assert np.array_equal(orc.calc_fir_coeffs(n, 0, 70, 4), ref.calc_fir_coeffs(n, 0, 70, 4))


def test_izero_msinc_match_reference_live(orc, ref):
    for x in np.linspace(0, 12, 97, dtype=np.float32):
        assert orc.lib.orc_izero(float(x)) == ref.lib.Izero(float(x))
    for m in range(-40, 41, 2):
        for fc in (0.01, 0.2333, 0.5):
            assert orc.lib.orc_m_sinc(m, fc) == ref.lib.m_sinc(m, fc)


# ---------------------------------------------------------------- A3/A4 q15 FIR -----------
FIR_TAPS = ["ssb_i", "ssb_q", "am102", "lp256", "lp512", "lp62", "wrap8", "n4", "n6"]


@pytest.mark.parametrize("tn", FIR_TAPS)
def test_fir_q15_matches_golden(orc, golden, tn):
    taps = golden["fir/taps_" + tn]
    for sn in ("noise", "full"):
        x = golden["fir/x_" + sn]
        for blk in (128, 130, 7):
            rc, y = orc.fir_q15_blocks(taps, x, blk)
            assert rc == 0
            assert np.array_equal(y, golden["fir/%s_%s_b%d" % (tn, sn, blk)]), (tn, sn, blk)


def test_fir_q15_accumulator_wraps_not_saturates(golden):
    # 8 taps of 32767 on full-scale input overflow 2^31: the golden (reference) output must differ
    # from a saturating/64-bit model, i.e. the fixture really exercises the wrap.
    x = golden["fir/x_full"].astype(np.int64)
    w = np.concatenate([np.zeros(7, np.int64), x])
    acc = np.array([(w[n:n + 8] * 32767).sum() for n in range(x.size)])
    wide = np.clip(acc >> 15, -32768, 32767)
    assert (wide != golden["fir/wrap8_full_b128"]).any()
    wrapped = ((acc + 2 ** 31) % 2 ** 32 - 2 ** 31) >> 15
    assert np.array_equal(np.clip(wrapped, -32768, 32767), golden["fir/wrap8_full_b128"])


def test_fir_q15_init_rejects_odd_taps(orc, golden):
    assert golden.meta["fir_init_odd_taps_status"] == -1      # ARM_MATH_ARGUMENT_ERROR
    rc, _ = orc.fir_q15_blocks(np.ones(5, np.int16), np.zeros(B, np.int16), B)
    assert rc == -1


def test_fir_q15_matches_reference_live(orc, ref):
    rng = np.random.default_rng(5)
    for _ in range(40):
        n = int(rng.integers(2, 140)) * 2
        amp = int(rng.choice([300, 8000, 32767]))
        taps = rng.integers(-amp, amp + 1, n).astype(np.int16)
        blk = int(rng.choice([1, 2, 3, 4, 5, 64, 127, 128, 129, 256]))
        x = rng.integers(-32768, 32768, blk * int(rng.integers(1, 5))).astype(np.int16)
        rc_o, y_o = orc.fir_q15_blocks(taps, x, blk)
        rc_r, y_r = ref.fir_q15_blocks(taps, x, blk)
        assert rc_o == rc_r == 0
        assert np.array_equal(y_o, y_r), (n, amp, blk)


def test_copy_and_sqrt_q31(orc, golden):
    assert np.array_equal(golden["copy_q15/out"], golden["chain/x_full"][:131])
    for v, want in zip(golden["sqrt_q31/in"], golden["sqrt_q31/out"]):
        assert orc.sqrt_q31(int(v))[1] == int(want), int(v)


def test_sqrt_q31_matches_reference_live(orc, ref):
    rng = np.random.default_rng(9)
    vals = list(rng.integers(-5, 2 ** 31, 3000)) + [2 ** 31 - 1, 1, 0, -2 ** 31] + [2 ** k for k in range(31)]
    for v in vals:
        assert orc.sqrt_q31(int(v)) == ref.sqrt_q31(int(v)), int(v)


# ---------------------------------------------------------------- A1+A4+A5 chain ----------
CHAIN = [("AM", orclib.AM, "fir/taps_am102", "fir/taps_am102"),
         ("LSB", orclib.LSB, "taps/FIR_SSB_I_coeffs", "taps/FIR_SSB_Q_coeffs"),
         ("USB", orclib.USB, "taps/FIR_SSB_I_coeffs", "taps/FIR_SSB_Q_coeffs"),
         ("CW", orclib.CW, "taps/FIR_CW_I_coeffs", "taps/FIR_CW_Q_coeffs")]


@pytest.mark.parametrize("sn", ["am", "tones", "noise", "full"])
@pytest.mark.parametrize("mn,mode,ti,tq", CHAIN)
def test_chain_q15_matches_golden(orc, golden, sn, mn, mode, ti, tq):
    x = golden["chain/x_" + sn]
    audio, i, q = orc.chain_q15(x, mode, golden[ti], golden[tq], want_iq=True)
    assert np.array_equal(i, golden["chain/%s_%s_I" % (sn, mn)])
    assert np.array_equal(q, golden["chain/%s_%s_Q" % (sn, mn)])
    assert np.array_equal(audio, golden["chain/%s_%s_audio" % (sn, mn)])
    if mn in ("AM", "CW"):
        a2 = orc.chain_q15(x, mode, golden[ti], golden[tq], sqrt_kind=orclib.SQRT_Q31)
        assert np.array_equal(a2, golden["chain/%s_%s_audio_q31" % (sn, mn)])


def test_static_tables_properties(golden):
    # Q = reverse(I) for both static pairs; sum|c| as probed
    for a, b, s in (("FIR_SSB_I_coeffs", "FIR_SSB_Q_coeffs", 59369), ("FIR_CW_I_coeffs", "FIR_CW_Q_coeffs", 42358)):
        assert np.array_equal(golden["taps/" + a], golden["taps/" + b][::-1])
        assert np.abs(golden["taps/" + a].astype(int)).sum() == s


# ---------------------------------------------------------------- row f4: spectrum FFT -----
def test_fft_tables_match_golden(orc, golden):
    """Generated twiddle / split-coefficient / bit-reversal tables == the reference's tables (data fixtures)."""
    tw, a, b = orc.fft_tables()
    assert np.array_equal(tw, golden["fft/twiddleCoef_64_q15"])
    assert np.array_equal(a, golden["fft/realCoefAQ15_stride64"])
    assert np.array_equal(b, golden["fft/realCoefBQ15_stride64"])
    pairs = lambda t: sorted((min(p), max(p)) for p in np.asarray(t, int).reshape(-1, 2).tolist())
    assert pairs(orc.bitrev_table64()) == pairs(golden["fft/armBitRevIndexTable_fixed_64"])


def test_rfft128_matches_golden(orc, golden):
    for x, want, work in zip(golden["fft/x"], golden["fft/rfft128_out"], golden["fft/rfft128_work"]):
        out, w = orc.rfft128_q15(x)
        assert np.array_equal(w, work)       # butterflies + bit reversal (what the reference leaves in `data`)
        assert np.array_equal(out, want)     # split stage


def test_rfft128_matches_reference_live(orc, ref):
    rng = np.random.default_rng(128)
    for case in range(300):
        amp = (300, 8000, 32767, 32767)[case % 4]
        x = rng.integers(-amp, amp + 1, 128).astype(np.int16)
        if case % 4 == 3:
            x[rng.integers(0, 128, 40)] = -32768
            x[rng.integers(0, 128, 40)] = 32767
        want, work = ref.rfft128_q15(x)
        out, w = orc.rfft128_q15(x)
        assert np.array_equal(w, work) and np.array_equal(out, want), case


def test_rfft128_is_a_dft(orc):
    """Independent of the reference: out[2k], out[2k+1] ~ Re, Im of (1/128) sum x[n] e^{-2 pi i k n / 128} (CMSIS scales
    the input down by 2 per radix-2 stage and the split halves once more) within the fixed-point noise."""
    rng = np.random.default_rng(5)
    x = rng.integers(-20000, 20001, 128).astype(np.int16)
    out, _ = orc.rfft128_q15(x)
    F = np.fft.fft(x.astype(float)) / 128
    got = out[0::2].astype(float) + 1j * out[1::2].astype(float)
    assert np.max(np.abs(got[1:64] - F[1:64])) < 6.0
    assert np.max(np.abs(got[65:] - F[65:])) < 6.0


def test_spectrum_columns_second_model(orc, golden):
    for out in golden["fft/rfft128_out"]:
        v = np.abs(out[127:0:-1].astype(np.int32)) // 200
        assert np.array_equal(orc.spectrum_columns(out), np.minimum(v, 16).astype(np.uint8))


def test_spectrum_tick_cadence(orc):
    import ctypes as C
    cnt = C.c_int(0)               # spectrumCounter starts at 0 -> first call draws, then every 25th
    hits = [i for i in range(80) if orc.lib.orc_spectrum_tick(1, C.byref(cnt))]
    assert hits == [0, 25, 50, 75]
    assert orc.lib.orc_spectrum_tick(0, C.byref(cnt)) == 0


# ---------------------------------------------------------------- A7 / f1: the arithmetic primitives ---------
# AudioFilterBiquad::update (filter_biquad.cpp:54-74) and the front end are built from the DSP-instruction wrappers of
# src/Audio/utility/dspinst.h.  The header carries them as Cortex-M4 inline assembly AND, for the Cortex-M0+, as plain C
# (`#elif defined(KINETISL)`): oracle/build_ref.sh compiles those C bodies as they stand.  The oracle's mulw16 / ssat16(v >> s) /
# history packing -- what its biquad, amplifier and DC blocker are made of -- must equal them on every input.
def _prims(orc):
    import ctypes as C
    L = orc.lib
    for n in ("orc_prim_mulw16b", "orc_prim_mulw16t"):
        getattr(L, n).restype = C.c_int32
        getattr(L, n).argtypes = [C.c_int32, C.c_uint32]
    L.orc_prim_ssat16_rshift.restype = C.c_int32
    L.orc_prim_ssat16_rshift.argtypes = [C.c_int32, C.c_int]
    L.orc_prim_pack_hist.restype = C.c_uint32
    L.orc_prim_pack_hist.argtypes = [C.c_int32, C.c_int32]
    return L


def test_dspinst_primitives_match_golden(orc):
    import hashlib
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(here, "dspinst_prims.npz"))
    want_sha = json.load(open(os.path.join(here, "dspinst_prims.sha256.json")))
    assert sorted(g.files) == sorted(want_sha)
    for k in g.files:
        assert hashlib.sha256(np.ascontiguousarray(g[k]).tobytes()).hexdigest() == want_sha[k], k
    L = _prims(orc)
    a, b = g["a"], g["b"]
    assert a.size > 20000
    for i in range(a.size):
        x, y = int(a[i]), int(b[i]) & 0xFFFFFFFF
        assert L.orc_prim_mulw16b(x, y) == int(g["mulwb"][i]), (x, y)
        assert L.orc_prim_mulw16t(x, y) == int(g["mulwt"][i]), (x, y)
        for sh in (0, 14, 15):
            assert L.orc_prim_ssat16_rshift(x, sh) == int(g["ssat16_asr%d" % sh][i]), (x, sh)
        # pack_16b_16b(a, b) = (a[15:0] << 16) | b[15:0]: the oracle packs (newer, older) history the same way
        assert L.orc_prim_pack_hist(x, int(b[i])) == int(g["pack_bb"][i]), (x, y)


def test_dspinst_primitives_match_reference_live(orc, ref):
    import ctypes as C
    L, R = _prims(orc), ref.lib
    for n in ("dspinst_signed_multiply_32x16b", "dspinst_signed_multiply_32x16t"):
        getattr(R, n).restype = C.c_int32
        getattr(R, n).argtypes = [C.c_int32, C.c_uint32]
    R.dspinst_signed_saturate_rshift.restype = C.c_int32
    R.dspinst_signed_saturate_rshift.argtypes = [C.c_int32, C.c_int, C.c_int]
    R.dspinst_pack_16b_16b.restype = C.c_uint32
    R.dspinst_pack_16b_16b.argtypes = [C.c_int32, C.c_int32]
    rng = np.random.default_rng(77)
    a = rng.integers(-2 ** 31, 2 ** 31, 30000).astype(np.int32)
    b = rng.integers(-2 ** 31, 2 ** 31, 30000).astype(np.int32)
    a[:8] = [0x7FFFFFFF, -0x80000000, 0x40000000, -0x40000000, -1, 1, 0x3FFF, -0x4000]
    b[:8] = [-0x80000000, 0x7FFFFFFF, 0x80008000 - (1 << 32), 0x7FFF7FFF, -1, 0x00010001, 0x8000, 0x7FFF]
    for x, y in zip(a, b):
        x, yu = int(x), int(y) & 0xFFFFFFFF
        assert L.orc_prim_mulw16b(x, yu) == R.dspinst_signed_multiply_32x16b(x, yu)
        assert L.orc_prim_mulw16t(x, yu) == R.dspinst_signed_multiply_32x16t(x, yu)
        for sh in (0, 1, 14, 15, 16):
            assert L.orc_prim_ssat16_rshift(x, sh) == R.dspinst_signed_saturate_rshift(x, 16, sh)
        assert L.orc_prim_pack_hist(x, int(y)) == R.dspinst_pack_16b_16b(x, int(y))


def test_am_sqrt_f32_matches_reference_live(orc, ref):
    """arm_sqrt_f32 (a static inline of the reference's arm_math.h, :5733-5758) as the Teensy-3.6 AM / CW branch calls it (Minimal-SDR.ino:611-612):
    every int32 the sum I*I + Q*Q can take goes through (float): negative sums (int overflow at full scale) give 0, not NaN."""
    import ctypes as C
    orc.lib.orc_prim_sqrt_f32.restype = C.c_float
    orc.lib.orc_prim_sqrt_f32.argtypes = [C.c_float]
    ref.lib.cmsis_arm_sqrt_f32.restype = C.c_int
    ref.lib.cmsis_arm_sqrt_f32.argtypes = [C.c_float, C.POINTER(C.c_float)]
    rng = np.random.default_rng(5)
    vals = np.concatenate([rng.integers(-2 ** 31, 2 ** 31, 20000).astype(np.float32), np.arange(0, 4096, dtype=np.float32) ** 2,
                           np.array([0.0, -0.0, 1e-45, -1e-45, 2147483648.0, -2147483648.0, 2.0 * 32768.0 ** 2, np.inf], np.float32)])
    for v in vals:
        out = C.c_float(-1.0)
        rc = ref.lib.cmsis_arm_sqrt_f32(float(v), C.byref(out))
        got = orc.lib.orc_prim_sqrt_f32(float(v))
        assert np.float32(got).tobytes() == np.float32(out.value).tobytes(), float(v)
        assert rc == (0 if v >= 0 else -1)
