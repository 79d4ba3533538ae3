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
