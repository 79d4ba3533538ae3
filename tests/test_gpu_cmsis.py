"""include/msdr_cmsis.h on the GPU: the reference's CMSIS-DSP argument lists (arm_fir_init_q15 / arm_fir_fast_q15,
arm_fir_init_f32 / arm_fir_f32, arm_biquad_cascade_df1_init_f32 / _f32; arm_math.h:1106-1128, 1182-1202, 1333-1351) over the
batched library, called through ctypes exactly as a relinked sketch would call them."""
import ctypes as C

import numpy as np
import pytest

import orclib
from gpuhelp import ctx, msdr, rel_rms  # noqa: F401

pytestmark = pytest.mark.gpu


class FirQ15(C.Structure):
    _fields_ = [("numTaps", C.c_uint16), ("pState", C.c_void_p), ("pCoeffs", C.c_void_p)]


class FirF32(C.Structure):
    _fields_ = [("numTaps", C.c_uint16), ("pState", C.c_void_p), ("pCoeffs", C.c_void_p)]


class BiquadF32(C.Structure):
    _fields_ = [("numStages", C.c_uint32), ("pState", C.c_void_p), ("pCoeffs", C.c_void_p)]


def test_arm_signatures_block_cadence(ctx, orc, golden):
    lib = ctx.lib
    lib.msdr_arm_fir_init_q15.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.msdr_arm_fir_init_f32.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.msdr_arm_biquad_cascade_df1_init_f32.argtypes = [C.c_void_p, C.c_uint8, C.c_void_p, C.c_void_p]
    for f in (lib.msdr_arm_fir_fast_q15, lib.msdr_arm_fir_f32, lib.msdr_arm_biquad_cascade_df1_f32):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        f.restype = None
    ch, B, blocks = 3, 128, 7
    assert lib.msdr_cmsis_bind(ctx.h, ch) == 0
    try:
        rng = np.random.default_rng(8)
        # --- Minimal-SDR.ino:906-927 / :574-575: init both FIR instances, then one call per block ---
        taps = golden["fir/taps_am102"]
        FIR_I, FIR_Q = FirQ15(), FirQ15()
        st_i, st_q = np.ones(102 + B, np.int16), np.ones(102 + B, np.int16)
        assert lib.msdr_arm_fir_init_q15(C.byref(FIR_I), 102, taps.ctypes.data, st_i.ctypes.data, B) == 0
        assert lib.msdr_arm_fir_init_q15(C.byref(FIR_Q), 102, taps.ctypes.data, st_q.ctypes.data, B) == 0
        assert (st_i == 0).all() and FIR_I.numTaps == 102 and FIR_I.pCoeffs == taps.ctypes.data
        x = rng.integers(-20000, 20001, (ch, blocks * B)).astype(np.int16)
        got = np.empty_like(x)
        for b in range(blocks):
            d_in, d_out = ctx.to_device(x[:, b * B:(b + 1) * B]), ctx.array((ch, B), np.int16)
            lib.msdr_arm_fir_fast_q15(C.byref(FIR_I), d_in.ptr, d_out.ptr, B)
            got[:, b * B:(b + 1) * B] = d_out.download()
        for c in range(ch):
            _, want = orc.fir_q15_blocks(taps, x[c], B)
            assert np.array_equal(got[c], want)
        # a re-init of the same instance starts a fresh filter (init_FIR on a retune)
        assert lib.msdr_arm_fir_init_q15(C.byref(FIR_I), 102, taps.ctypes.data, st_i.ctypes.data, B) == 0
        d_in, d_out = ctx.to_device(x[:, :B]), ctx.array((ch, B), np.int16)
        lib.msdr_arm_fir_fast_q15(C.byref(FIR_I), d_in.ptr, d_out.ptr, B)
        assert np.array_equal(d_out.download(), got[:, :B])
        # --- arm_fir_f32 and arm_biquad_cascade_df1_f32 ---
        h = (rng.standard_normal(61) / 8).astype(np.float32)
        Sf, stf = FirF32(), np.ones(61 + B - 1, np.float32)
        lib.msdr_arm_fir_init_f32(C.byref(Sf), 61, h.ctypes.data, stf.ctypes.data, B)
        assert (stf == 0).all()
        bq = np.array([[0.2066, 0.4131, 0.2066, 0.3695, -0.1958], [0.9766, -1.3815, 0.9766, 1.3815, -0.9533]], np.float32)
        Sb, stb = BiquadF32(), np.ones(8, np.float32)
        lib.msdr_arm_biquad_cascade_df1_init_f32(C.byref(Sb), 2, bq.ctypes.data, stb.ctypes.data)
        assert (stb == 0).all() and Sb.numStages == 2
        xf = rng.uniform(-1, 1, (ch, blocks * B)).astype(np.float32)
        gf = np.empty_like(xf)
        for b in range(blocks):
            d_in, d_mid, d_out = ctx.to_device(xf[:, b * B:(b + 1) * B]), ctx.array((ch, B), np.float32), ctx.array((ch, B), np.float32)
            lib.msdr_arm_fir_f32(C.byref(Sf), d_in.ptr, d_mid.ptr, B)
            lib.msdr_arm_biquad_cascade_df1_f32(C.byref(Sb), d_mid.ptr, d_out.ptr, B)
            gf[:, b * B:(b + 1) * B] = d_out.download()
        for c in range(ch):
            want = orc.biquad_df1_blocks(bq, orc.fir_f32_blocks(h, xf[c], B), B)
            assert rel_rms(gf[c], want) < 2e-6
        # odd tap counts are refused like the reference refuses them
        assert lib.msdr_arm_fir_init_q15(C.byref(FIR_Q), 101, taps.ctypes.data, st_q.ctypes.data, B) == -1
    finally:
        assert lib.msdr_cmsis_bind(None, 0) == 0


def _sigs(lib):
    lib.msdr_arm_fir_init_q15.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.msdr_arm_fir_init_f32.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.msdr_arm_biquad_cascade_df1_init_f32.argtypes = [C.c_void_p, C.c_uint8, C.c_void_p, C.c_void_p]
    lib.msdr_cmsis_bind_host.argtypes = [C.c_void_p, C.c_uint32]
    for f in (lib.msdr_arm_fir_fast_q15, lib.msdr_arm_fir_f32, lib.msdr_arm_biquad_cascade_df1_f32):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        f.restype = None


@pytest.mark.parametrize("ch", [1, 5])
def test_arm_signatures_on_host_arrays_as_the_sketch_passes_them(ctx, orc, golden, ch):
    """msdr_cmsis_bind_host: pSrc / pDst are the caller's HOST arrays -- `arm_fir_fast_q15(&FIR_I, I_buffer, I_FIR_out, AUDIO_BLOCK_SAMPLES)`
    on stack buffers (Minimal-SDR.ino:525-526, 574-575); channels = 1 is exactly the sketch's call.  Bit-exact against the reference-generated
    golden FIR vectors, state kept across an in-place rewrite of pCoeffs (the bandwidth menu, UI.cpp:337-345)."""
    lib = ctx.lib
    _sigs(lib)
    B = 128
    assert lib.msdr_cmsis_bind_host(ctx.h, ch) == 0
    try:
        # --- the golden known answers of arm_fir_fast_q15 (the compiled reference's own outputs), one 128-sample call per block on numpy arrays ---
        FIR_I = FirQ15()
        st = np.ones(512 + B, np.int16)
        if ch == 1:
            for tn, tk in (("am102", "fir/taps_am102"), ("ssb_i", "fir/taps_ssb_i"), ("lp256", "fir/taps_lp256")):
                tg = golden[tk].copy()
                for sn in ("noise", "full"):
                    xg, yg = golden["fir/x_" + sn], golden["fir/%s_%s_b128" % (tn, sn)]
                    assert lib.msdr_arm_fir_init_q15(C.byref(FIR_I), tg.size, tg.ctypes.data, st.ctypes.data, B) == 0
                    src, dst, gy = np.empty(B, np.int16), np.empty(B, np.int16), np.empty_like(yg)
                    for o in range(0, xg.size, B):
                        src[:] = xg[o:o + B]
                        lib.msdr_arm_fir_fast_q15(C.byref(FIR_I), src.ctypes.data, dst.ctypes.data, B)
                        gy[o:o + B] = dst
                    assert np.array_equal(gy, yg), (tn, sn)
        # --- state kept across an in-place rewrite of pCoeffs: calc_demod_filter() rewrites FIR_AM_coeffs, no init_FIR() ---
        taps = golden["fir/taps_am102"].copy()
        t0 = golden["fir/taps_am102"]
        assert lib.msdr_arm_fir_init_q15(C.byref(FIR_I), 102, taps.ctypes.data, st.ctypes.data, B) == 0
        rng = np.random.default_rng(80 + ch)
        blocks = 9
        x = rng.integers(-32768, 32768, (ch, blocks * B)).astype(np.int16)
        got = np.empty_like(x)
        I_buffer, I_FIR_out = np.empty((ch, B), np.int16), np.empty((ch, B), np.int16)     # the sketch's two arrays
        taps2 = orc.calc_fir_coeffs(102, 1500.0)[:102].copy()
        for b in range(blocks):
            if b == 5:
                taps[:] = taps2
            I_buffer[:] = x[:, b * B:(b + 1) * B]
            lib.msdr_arm_fir_fast_q15(C.byref(FIR_I), I_buffer.ctypes.data, I_FIR_out.ctypes.data, B)
            got[:, b * B:(b + 1) * B] = I_FIR_out
        for c in range(ch):
            # a FIR's state is its last samples: after the rewrite the output equals the whole stream filtered with the new taps
            _, a_ = orc.fir_q15_blocks(t0, x[c], B)
            _, b_ = orc.fir_q15_blocks(taps2, x[c], B)
            assert np.array_equal(got[c], np.concatenate([a_[:5 * B], b_[5 * B:]])), c
        # --- fp32: arm_fir_f32 then arm_biquad_cascade_df1_f32 IN PLACE on one host array ---
        h = (rng.standard_normal(61) / 8).astype(np.float32)
        Sf, stf = FirF32(), np.ones(61 + B - 1, np.float32)
        lib.msdr_arm_fir_init_f32(C.byref(Sf), 61, h.ctypes.data, stf.ctypes.data, B)
        bq = np.array([[0.2066, 0.4131, 0.2066, 0.3695, -0.1958], [0.9766, -1.3815, 0.9766, 1.3815, -0.9533]], np.float32)
        Sb, stb = BiquadF32(), np.ones(8, np.float32)
        lib.msdr_arm_biquad_cascade_df1_init_f32(C.byref(Sb), 2, bq.ctypes.data, stb.ctypes.data)
        xf = rng.uniform(-1, 1, (ch, 6 * B)).astype(np.float32)
        gf = np.empty_like(xf)
        buf_in, buf = np.empty((ch, B), np.float32), np.empty((ch, B), np.float32)
        for b in range(6):
            buf_in[:] = xf[:, b * B:(b + 1) * B]
            lib.msdr_arm_fir_f32(C.byref(Sf), buf_in.ctypes.data, buf.ctypes.data, B)
            lib.msdr_arm_biquad_cascade_df1_f32(C.byref(Sb), buf.ctypes.data, buf.ctypes.data, B)      # pSrc == pDst
            gf[:, b * B:(b + 1) * B] = buf
        for c in range(ch):
            want = orc.biquad_df1_blocks(bq, orc.fir_f32_blocks(h, xf[c], B), B)
            assert rel_rms(gf[c], want) < 2e-6
    finally:
        assert lib.msdr_cmsis_bind(None, 0) == 0


def test_a_failed_coefficient_rebuild_keeps_the_old_tables_answering(ctx, orc):
    """the shim took the caller's new pCoeffs into its snapshot BEFORE the rebuild had succeeded and returned without writing
    pDst when it failed.  The one rebuild that can fail on content is the cascade's: its state is carried through the OLD coefficients'
    basis, which is singular where a later numerator cancels an earlier denominator (msdr_biquad_df1_f32_set_coeffs -> ARGUMENT_ERROR).
    Now: the block still runs, on the tables the filter has; the snapshot keeps the bytes those tables were built from (every later call
    compares unequal and tries again); a re-init -- the caller's way out, as init_FIR() on a retune -- starts a fresh filter."""
    lib = ctx.lib
    _sigs(lib)
    lib.msdr_last_error.restype = C.c_char_p
    B, ch = 128, 2
    assert lib.msdr_cmsis_bind_host(ctx.h, ch) == 0
    try:
        good = np.array([[0.2066, 0.4131, 0.2066, 0.3695, -0.1958], [0.9766, -1.3815, 0.9766, 1.3815, -0.9533]], np.float32)
        bad = good.copy()
        bad[1, :3] = [1.0, -good[0, 3], -good[0, 4]]             # section 1's numerator = section 0's denominator (CMSIS sign: 1 - a1 z^-1 - a2 z^-2)
        new = good.copy()
        new[0] = [0.3, 0.6, 0.3, 0.2, -0.3]
        coeffs = good.copy()
        Sb, stb = BiquadF32(), np.zeros(8, np.float32)
        lib.msdr_arm_biquad_cascade_df1_init_f32(C.byref(Sb), 2, coeffs.ctypes.data, stb.ctypes.data)
        rng = np.random.default_rng(91)
        x = rng.uniform(-1, 1, (ch, 5 * B)).astype(np.float32)
        out = np.full((ch, 5 * B), np.nan, np.float32)
        buf_in, buf_out = np.empty((ch, B), np.float32), np.empty((ch, B), np.float32)

        def call(b):
            buf_in[:] = x[:, b * B:(b + 1) * B]
            buf_out[:] = np.nan
            lib.msdr_arm_biquad_cascade_df1_f32(C.byref(Sb), buf_in.ctypes.data, buf_out.ctypes.data, B)
            out[:, b * B:(b + 1) * B] = buf_out

        call(0)                                                    # good
        coeffs[:] = bad
        call(1)                                                    # good -> bad: the state is read in good's basis: succeeds
        coeffs[:] = new
        call(2)                                                    # bad -> new: the state would be read in bad's basis: refused; the block runs on bad's tables
        assert b"" != lib.msdr_last_error()
        assert np.isfinite(out[:, 2 * B:3 * B]).all()
        call(3)                                                    # tried again (same refusal), still bad's tables
        # the oracle does what CMSIS does: one instance, its coefficient array rewritten in place -- here good, then bad from block 1 on
        for c in range(ch):
            cf = np.ascontiguousarray(good, np.float32).reshape(-1).copy()
            S, st_o = orclib.BiquadDf1(), np.zeros(8, np.float32)
            orc.lib.orc_biquad_df1_init_f32(C.byref(S), C.c_uint8(2), orclib._ptr(cf), orclib._ptr(st_o))
            y = np.empty(4 * B, np.float32)
            xs = np.ascontiguousarray(x[c])
            for k in range(4):
                if k == 1:
                    cf[:] = bad.reshape(-1)
                orc.lib.orc_biquad_df1_f32_run(C.byref(S), orclib._ptr(xs[k * B:(k + 1) * B]), orclib._ptr(y[k * B:(k + 1) * B]), C.c_uint32(B))
            assert rel_rms(out[c, :4 * B], y) < 5e-6, (c, rel_rms(out[c, :4 * B], y))
        # the caller's way out: init again with the coefficients it wants
        lib.msdr_arm_biquad_cascade_df1_init_f32(C.byref(Sb), 2, coeffs.ctypes.data, stb.ctypes.data)
        call(4)
        for c in range(ch):
            assert rel_rms(out[c, 4 * B:], orc.biquad_df1_blocks(new, x[c, 4 * B:], B)) < 2e-6, c
    finally:
        assert lib.msdr_cmsis_bind(None, 0) == 0
