"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, where present, the compiled
reference (oracle/_ref/libmsdr_ref.so).  TEST INFRASTRUCTURE: imported by tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() only -- never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libmsdr_ref.so")

SYNCAM, AM, LSB, USB, CW = 0, 1, 2, 3, 4          # stations.h:4
SQRT_F32, SQRT_Q31 = 0, 1
BLOCK = 128
AUDIO_SAMPLE_RATE_EXACT = 44117.64706
BQ_LOWPASS, BQ_HIGHPASS, BQ_BANDPASS, BQ_NOTCH, BQ_LOWSHELF, BQ_HIGHSHELF = range(6)

_p = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(_p) if a is not None else None


def build_oracle(force=False):
    if force or not os.path.exists(ORACLE_SO) or (
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "msdr_oracle.c"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


class FirQ15(C.Structure):
    _fields_ = [("numTaps", C.c_uint16), ("pState", _p), ("pCoeffs", _p)]


class FirF32(C.Structure):
    _fields_ = [("numTaps", C.c_uint16), ("pState", _p), ("pCoeffs", _p)]


class BiquadTeensy(C.Structure):
    _fields_ = [("definition", C.c_int32 * 32)]


class BiquadDf1(C.Structure):
    _fields_ = [("numStages", C.c_uint32), ("pState", _p), ("pCoeffs", _p)]


class ChainQ15Cfg(C.Structure):
    _fields_ = [("mode", C.c_int), ("sqrt_kind", C.c_int), ("mixer", C.c_int),
                ("num_taps", C.c_uint32), ("coeffs_i", _p), ("coeffs_q", _p),
                ("osc_i", _p), ("osc_q", _p), ("n_biquad_nodes", C.c_uint32), ("bq_init", _p)]


class ChainQ15State(C.Structure):
    _fields_ = [("state_i", _p), ("state_q", _p), ("bq", BiquadTeensy * 2)]


class ChainF32Cfg(C.Structure):
    _fields_ = [("mode", C.c_int), ("in_scale", C.c_float), ("num_taps", C.c_uint32),
                ("coeffs_i", _p), ("coeffs_q", _p), ("osc_len", C.c_uint32),
                ("osc_i", _p), ("osc_q", _p), ("num_stages", C.c_uint32), ("bq_coeffs", _p)]


class ChainF32State(C.Structure):
    _fields_ = [("hist_i", _p), ("hist_q", _p), ("bq_state", C.c_float * 16), ("n0", C.c_uint64)]


class DcBlock(C.Structure):
    _fields_ = [("hpf_y1", C.c_int32), ("hpf_x1", C.c_int32)]


class Agc(C.Structure):
    _fields_ = [("agc_buffer", C.c_int16 * 25), ("agc_idx", C.c_int32), ("AGC_val", C.c_float), ("AGC_on", C.c_int32),
                ("multiplier", C.c_int32)]


class Syncam(C.Structure):
    _fields_ = [("fil_out", C.c_float), ("omega2", C.c_float), ("phzerror", C.c_float)]


class Anr(C.Structure):
    _fields_ = [("lidx", C.c_float), ("ngamma", C.c_float), ("in_idx", C.c_int32), ("d", C.c_float * 512), ("w", C.c_float * 512)]


class Frontend(C.Structure):
    _fields_ = [("dc", DcBlock), ("agc", Agc)]


class ChainF32Post(C.Structure):
    _fields_ = [("pll", C.c_int32), ("anr_on", C.c_int32), ("pll_state", Syncam), ("anr_state", Anr)]


class Oracle:
    """Thin, numpy-in/numpy-out face of liboracle.so."""

    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        L.orc_izero.restype = C.c_float
        L.orc_izero.argtypes = [C.c_float]
        L.orc_m_sinc.restype = C.c_float
        L.orc_m_sinc.argtypes = [C.c_int, C.c_float]
        L.orc_calc_fir_coeffs.argtypes = [_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
        L.orc_biquad_design.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_double, _p]
        L.orc_sqrt_q31.argtypes = [C.c_int32, _p]
        L.orc_chain_f32.argtypes = [_p, _p, _p, _p, C.c_uint64]
        L.orc_chain_f32_batch.argtypes = [_p, _p, _p, _p, C.c_uint32, C.c_uint64, C.c_int]
        L.orc_chain_q15_batch.argtypes = [_p, _p, _p, _p, C.c_uint32, C.c_uint32, C.c_int]

    # ---- A1 / A2 ------------------------------------------------------------------------
    def mix_fs4(self, x):
        x = np.ascontiguousarray(x, np.int16)
        i = np.empty_like(x)
        q = np.empty_like(x)
        self.lib.orc_mix_fs4_q15(_ptr(x), _ptr(i), _ptr(q), C.c_uint32(x.size))
        return i, q

    def freqconv_q15(self, i, q, osc_i, osc_q, direction, passthrough):
        i = np.array(i, np.int16)
        q = np.array(q, np.int16)
        oi = np.ascontiguousarray(osc_i, np.int16)
        oq = np.ascontiguousarray(osc_q, np.int16)
        self.lib.orc_freqconv_q15(_ptr(i), _ptr(q), _ptr(oi), _ptr(oq), int(direction), int(passthrough),
                                  C.c_uint32(i.size))
        return i, q

    def freqconv_f32(self, i, q, osc_i, osc_q, direction, passthrough):
        i = np.array(i, np.float32)
        q = np.array(q, np.float32)
        oi = np.ascontiguousarray(osc_i, np.float32)
        oq = np.ascontiguousarray(osc_q, np.float32)
        self.lib.orc_freqconv_f32(_ptr(i), _ptr(q), _ptr(oi), _ptr(oq), int(direction), int(passthrough),
                                  C.c_uint32(i.size))
        return i, q

    # ---- A3 / A4 ------------------------------------------------------------------------
    def fir_q15_blocks(self, coeffs, x, block):
        """Run consecutive blocks through one instance (state carried); returns (status, y)."""
        coeffs = np.ascontiguousarray(coeffs, np.int16)
        x = np.ascontiguousarray(x, np.int16)
        st = np.full(coeffs.size + block, 0x5A5A, np.int16)   # init must clear it
        S = FirQ15()
        rc = self.lib.orc_fir_init_q15(C.byref(S), C.c_uint16(coeffs.size), _ptr(coeffs), _ptr(st),
                                       C.c_uint32(block))
        if rc != 0:
            return rc, None
        y = np.empty_like(x)
        for o in range(0, x.size, block):
            n = min(block, x.size - o)
            self.lib.orc_fir_fast_q15(C.byref(S), _ptr(x[o:o + n]), _ptr(y[o:o + n]), C.c_uint32(n))
        return 0, y

    def fir_f32_blocks(self, coeffs, x, block):
        coeffs = np.ascontiguousarray(coeffs, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        st = np.full(coeffs.size + block, 7.0, np.float32)
        S = FirF32()
        self.lib.orc_fir_init_f32(C.byref(S), C.c_uint16(coeffs.size), _ptr(coeffs), _ptr(st), C.c_uint32(block))
        y = np.empty_like(x)
        for o in range(0, x.size, block):
            n = min(block, x.size - o)
            self.lib.orc_fir_f32(C.byref(S), _ptr(x[o:o + n]), _ptr(y[o:o + n]), C.c_uint32(n))
        return y

    # ---- A5 -----------------------------------------------------------------------------
    def sqrt_q31(self, v):
        out = C.c_int32(0)
        rc = self.lib.orc_sqrt_q31(C.c_int32(int(v)), C.byref(out))
        return rc, out.value

    def demod_q15(self, mode, i, q, sqrt_kind=SQRT_F32):
        i = np.ascontiguousarray(i, np.int16)
        q = np.ascontiguousarray(q, np.int16)
        out = np.empty_like(i)
        self.lib.orc_demod_q15(int(mode), int(sqrt_kind), _ptr(i), _ptr(q), _ptr(out), C.c_uint32(i.size))
        return out

    def demod_f32(self, mode, i, q):
        i = np.ascontiguousarray(i, np.float32)
        q = np.ascontiguousarray(q, np.float32)
        out = np.empty_like(i)
        self.lib.orc_demod_f32(int(mode), _ptr(i), _ptr(q), _ptr(out), C.c_uint32(i.size))
        return out

    # ---- A7 -----------------------------------------------------------------------------
    def biquad_design(self, kind, freq, q_or_gain, slope=1.0, fs=AUDIO_SAMPLE_RATE_EXACT):
        c = np.zeros(5, np.int32)
        self.lib.orc_biquad_design(int(kind), float(freq), float(q_or_gain), float(slope), float(fs), _ptr(c))
        return c

    def biquad_teensy_new(self, stage_coefs):
        b = BiquadTeensy()
        self.lib.orc_biquad_teensy_init(C.byref(b))
        for s, c in enumerate(stage_coefs):
            c = np.ascontiguousarray(c, np.int32)
            self.lib.orc_biquad_teensy_set_coefficients(C.byref(b), C.c_uint32(s), _ptr(c))
        return b

    def biquad_teensy_update(self, b, data):
        d = np.array(data, np.int16)
        self.lib.orc_biquad_teensy_update(C.byref(b), _ptr(d), C.c_uint32(d.size))
        return d

    # ---- A8 -----------------------------------------------------------------------------
    def biquad_df1_blocks(self, coeffs, x, block):
        coeffs = np.ascontiguousarray(coeffs, np.float32).reshape(-1)
        ns = coeffs.size // 5
        x = np.ascontiguousarray(x, np.float32)
        st = np.full(4 * max(ns, 1), 3.0, np.float32)
        S = BiquadDf1()
        self.lib.orc_biquad_df1_init_f32(C.byref(S), C.c_uint8(ns), _ptr(coeffs), _ptr(st))
        y = np.empty_like(x)
        for o in range(0, x.size, block):
            n = min(block, x.size - o)
            self.lib.orc_biquad_df1_f32_run(C.byref(S), _ptr(x[o:o + n]), _ptr(y[o:o + n]), C.c_uint32(n))
        return y

    def biquad_df1_zero_state(self, coeffs, x):
        """arm_biquad_cascade_df1_f32 over x from zero state (one call)"""
        coeffs = np.ascontiguousarray(coeffs, np.float32).reshape(-1)
        ns = coeffs.size // 5
        x = np.ascontiguousarray(x, np.float32)
        st = np.zeros(4 * max(ns, 1), np.float32)
        S = BiquadDf1()
        self.lib.orc_biquad_df1_init_f32(C.byref(S), C.c_uint8(ns), _ptr(coeffs), _ptr(st))
        y = np.empty_like(x)
        self.lib.orc_biquad_df1_f32_run(C.byref(S), _ptr(x), _ptr(y), C.c_uint32(x.size))
        return y

    # ---- A9 -----------------------------------------------------------------------------
    # ---- row f3: LMS automatic notch / noise reduction ----
    def anr_new(self):
        a = Anr()
        self.lib.orc_anr_init(C.byref(a))
        return a

    def anr_q15(self, a, anr_on, data):
        d = np.array(data, np.int16)
        self.lib.orc_anr_q15(C.byref(a), C.c_int(int(anr_on)), _ptr(d), C.c_uint32(d.size))
        return d

    # ---- row f2: SYNCAM PLL ----
    def syncam_new(self):
        s = Syncam()
        self.lib.orc_syncam_init(C.byref(s))
        return s

    def syncam_q15(self, s, i, q):
        i, q = np.ascontiguousarray(i, np.int16), np.ascontiguousarray(q, np.int16)
        out = np.empty(i.size, np.int16)
        self.lib.orc_syncam_q15(C.byref(s), _ptr(i), _ptr(q), _ptr(out), C.c_uint32(i.size))
        return out

    def syncam_f32(self, s, i, q):
        i, q = np.ascontiguousarray(i, np.float32), np.ascontiguousarray(q, np.float32)
        out = np.empty(i.size, np.float32)
        self.lib.orc_syncam_f32(C.byref(s), _ptr(i), _ptr(q), _ptr(out), C.c_uint32(i.size))
        return out

    def anr_f32(self, a, anr_on, data):
        d = np.array(data, np.float32)
        self.lib.orc_anr_f32(C.byref(a), C.c_int(int(anr_on)), _ptr(d), C.c_uint32(d.size))
        return d

    def syncam_constants(self):
        c = np.zeros(4, np.float32)
        self.lib.orc_syncam_constants(_ptr(c))
        return c

    # ---- row f1: front end (DC block, AudioAmplifier, AGC) ----
    def frontend_new(self, first_conversion=0, agc_on=True, gain=None):
        f = Frontend()
        self.lib.orc_frontend_init(C.byref(f), C.c_uint16(int(first_conversion)))
        f.agc.AGC_on = 1 if agc_on else 0
        if gain is not None:
            f.agc.AGC_val = gain
            f.agc.multiplier = self.amp_multiplier(gain)
        return f

    def amp_multiplier(self, gain):
        self.lib.orc_amp_multiplier.restype = C.c_int32
        self.lib.orc_amp_multiplier.argtypes = [C.c_float]
        return int(self.lib.orc_amp_multiplier(C.c_float(gain)))

    def frontend_run(self, f, adc):
        adc = np.ascontiguousarray(adc, np.uint16)
        assert adc.size % BLOCK == 0
        out = np.empty(adc.size, np.int16)
        self.lib.orc_frontend_run(C.byref(f), _ptr(adc), _ptr(out), C.c_uint32(adc.size // BLOCK))
        return out

    def dcblock(self, st, adc):
        adc = np.ascontiguousarray(adc, np.uint16)
        out = np.empty(adc.size, np.int16)
        self.lib.orc_dcblock_update(C.byref(st), _ptr(adc), _ptr(out), C.c_uint32(adc.size))
        return out

    def amp_update(self, mult, data):
        d = np.ascontiguousarray(data, np.int16).copy()
        self.lib.orc_amp_update.restype = C.c_int
        ok = self.lib.orc_amp_update(C.c_int32(int(mult)), _ptr(d), C.c_uint32(d.size))
        return (d if ok else None)

    def agc_block(self, a, block):
        b = np.ascontiguousarray(block, np.int16)
        assert b.size == BLOCK
        self.lib.orc_agc_block(C.byref(a), _ptr(b))

    def fft_tables(self):
        tw, a, b = np.zeros(96, np.int16), np.zeros(128, np.int16), np.zeros(128, np.int16)
        self.lib.orc_fft_tables(_ptr(tw), _ptr(a), _ptr(b))
        return tw, a, b

    def bitrev_table64(self):
        t = np.zeros(56, np.uint16)
        n = self.lib.orc_bitrev_table64(_ptr(t))
        return t[:n]

    def rfft128_q15(self, x):
        x = np.ascontiguousarray(x, np.int16)
        out, work = np.zeros(256, np.int16), np.zeros(128, np.int16)
        self.lib.orc_rfft128_q15(_ptr(x), _ptr(out), _ptr(work))
        return out, work

    def spectrum_columns(self, fft_out):
        fft_out = np.ascontiguousarray(fft_out, np.int16)
        y = np.zeros(127, np.uint8)
        self.lib.orc_spectrum_columns(_ptr(fft_out), _ptr(y))
        return y

    def calc_fir_coeffs(self, n, fc, astop=70.0, ftype=0, dfc=0.0, fs=24000.0, pi_double=False, room=None):
        self.lib.orc_set_pi_double(int(pi_double))
        buf = np.zeros(room or (2 * n + 8), np.int16)
        self.lib.orc_calc_fir_coeffs(_ptr(buf), int(n), float(fc), float(astop), int(ftype), float(dfc), float(fs))
        self.lib.orc_set_pi_double(0)
        return buf

    # ---- chains -------------------------------------------------------------------------
    def chain_q15(self, x, mode, coeffs_i, coeffs_q, mixer=0, osc_i=None, osc_q=None,
                  sqrt_kind=SQRT_F32, biquads=(), want_iq=False, state=None):
        """One channel, len(x) a multiple of BLOCK. biquads = list of BiquadTeensy nodes (copied).
        `state` = dict carried between calls (FIR states, biquad nodes): the stream continues where the last call stopped."""
        x = np.ascontiguousarray(x, np.int16)
        assert x.size % BLOCK == 0
        ci = np.ascontiguousarray(coeffs_i, np.int16)
        cq = np.ascontiguousarray(coeffs_q, np.int16)
        oi = np.ascontiguousarray(osc_i, np.int16) if osc_i is not None else None
        oq = np.ascontiguousarray(osc_q, np.int16) if osc_q is not None else None
        cfg = ChainQ15Cfg(int(mode), int(sqrt_kind), int(mixer), ci.size, _ptr(ci), _ptr(cq),
                          _ptr(oi), _ptr(oq), len(biquads), None)
        if state is None:
            state = {}
        si = state.setdefault("si", np.zeros(ci.size + BLOCK, np.int16))
        sq = state.setdefault("sq", np.zeros(ci.size + BLOCK, np.int16))
        st = ChainQ15State()
        st.state_i, st.state_q = _ptr(si), _ptr(sq)
        for k, b in enumerate(state.get("bq", biquads)):
            C.memmove(C.byref(st.bq[k]), C.byref(b), C.sizeof(BiquadTeensy))
        audio = np.empty_like(x)
        io = np.empty_like(x) if want_iq else None
        qo = np.empty_like(x) if want_iq else None
        self.lib.orc_chain_q15(C.byref(cfg), C.byref(st), _ptr(x), _ptr(audio), _ptr(io), _ptr(qo),
                               C.c_uint32(x.size // BLOCK))
        if biquads:
            keep = []
            for k in range(len(biquads)):
                b = BiquadTeensy()
                C.memmove(C.byref(b), C.byref(st.bq[k]), C.sizeof(BiquadTeensy))
                keep.append(b)
            state["bq"] = keep
        return (audio, io, qo) if want_iq else audio

    def _f32_cfg(self, mode, coeffs_i, coeffs_q, osc_i, osc_q, bq_coeffs, in_scale):
        keep = [np.ascontiguousarray(coeffs_i, np.float32), np.ascontiguousarray(coeffs_q, np.float32),
                np.ascontiguousarray(osc_i, np.float32), np.ascontiguousarray(osc_q, np.float32),
                np.ascontiguousarray(bq_coeffs if bq_coeffs is not None else [], np.float32).reshape(-1)]
        cfg = ChainF32Cfg(int(mode), float(in_scale), keep[0].size, _ptr(keep[0]), _ptr(keep[1]),
                          keep[2].size, _ptr(keep[2]), _ptr(keep[3]), keep[4].size // 5,
                          _ptr(keep[4]) if keep[4].size else None)
        return cfg, keep

    def chain_f32(self, x, mode, coeffs_i, coeffs_q, osc_i, osc_q, bq_coeffs=None, in_scale=1.0 / 32768,
                  state=None, pll=False, anr_on=0):
        """One channel; `state` = dict carried between calls (hist_i, hist_q, bq, n0, post).
        pll: a SYNCAM channel demodulates through the PLL (.ino:631-688); anr_on 1 / 2: the LMS filter (.ino:702-770) between
        demodulator and cascade -- the fp32 flavours of rows f2 / f3 (oracle/msdr_oracle.h)."""
        x = np.ascontiguousarray(x, np.int16)
        cfg, keep = self._f32_cfg(mode, coeffs_i, coeffs_q, osc_i, osc_q, bq_coeffs, in_scale)
        h = max(keep[0].size - 1, 1)
        if state is None:
            state = {}
        hi = state.setdefault("hist_i", np.zeros(h, np.float32))
        hq = state.setdefault("hist_q", np.zeros(h, np.float32))
        st = ChainF32State()
        st.hist_i, st.hist_q = _ptr(hi), _ptr(hq)
        for k, v in enumerate(state.get("bq", np.zeros(16, np.float32))):
            st.bq_state[k] = float(v)
        st.n0 = int(state.get("n0", 0))
        out = np.empty(x.size, np.float32)
        if pll or anr_on:
            post = state.get("post")
            if post is None:
                post = ChainF32Post()
                self.lib.orc_chain_f32_post_init(C.byref(post), C.c_int(1 if pll else 0), C.c_int(int(anr_on)))
                state["post"] = post
            self.lib.orc_chain_f32_post_run(C.byref(cfg), C.byref(st), C.byref(post), _ptr(x), _ptr(out), C.c_uint64(x.size))
        else:
            self.lib.orc_chain_f32(C.byref(cfg), C.byref(st), _ptr(x), _ptr(out), C.c_uint64(x.size))
        state["bq"] = np.array(list(st.bq_state), np.float32)
        state["n0"] = int(st.n0)
        return out

    def chain_f32_batch(self, x, modes, coeffs_i, coeffs_q, osc_i, osc_q, bq_coeffs=None,
                        in_scale=1.0 / 32768, threads=0):
        """x: [channels, n] int16; modes: int or per-channel array. Fresh zero state per channel."""
        x = np.ascontiguousarray(x, np.int16)
        ch, n = x.shape
        mode0 = int(modes) if np.isscalar(modes) else AM
        mp = None if np.isscalar(modes) else np.ascontiguousarray(modes, np.int32)
        cfg, keep = self._f32_cfg(mode0, coeffs_i, coeffs_q, osc_i, osc_q, bq_coeffs, in_scale)
        out = np.empty((ch, n), np.float32)
        used = self.lib.orc_chain_f32_batch(C.byref(cfg), _ptr(mp), _ptr(x), _ptr(out), C.c_uint32(ch),
                                            C.c_uint64(n), int(threads))
        return out, used

    def chain_q15_batch(self, x, modes, coeffs_i, coeffs_q, mixer=0, osc_i=None, osc_q=None,
                        sqrt_kind=SQRT_F32, biquads=(), threads=0):
        x = np.ascontiguousarray(x, np.int16)
        ch, n = x.shape
        assert n % BLOCK == 0
        ci = np.ascontiguousarray(coeffs_i, np.int16)
        cq = np.ascontiguousarray(coeffs_q, np.int16)
        oi = np.ascontiguousarray(osc_i, np.int16) if osc_i is not None else None
        oq = np.ascontiguousarray(osc_q, np.int16) if osc_q is not None else None
        bq = (BiquadTeensy * 2)()
        for k, b in enumerate(biquads):
            C.memmove(C.byref(bq[k]), C.byref(b), C.sizeof(BiquadTeensy))
        mode0 = int(modes) if np.isscalar(modes) else AM
        mp = None if np.isscalar(modes) else np.ascontiguousarray(modes, np.int32)
        cfg = ChainQ15Cfg(mode0, int(sqrt_kind), int(mixer), ci.size, _ptr(ci), _ptr(cq), _ptr(oi), _ptr(oq),
                          len(biquads), C.cast(bq, _p))
        out = np.empty_like(x)
        used = self.lib.orc_chain_q15_batch(C.byref(cfg), _ptr(mp), _ptr(x), _ptr(out), C.c_uint32(ch),
                                            C.c_uint32(n // BLOCK), int(threads))
        return out, used


class ArmCfftRadix4Q15(C.Structure):   # arm_math.h:1963-1972 (arm_cfft_radix4_instance_q15)
    _fields_ = [("fftLen", C.c_uint16), ("ifftFlag", C.c_uint8), ("bitReverseFlag", C.c_uint8), ("pTwiddle", _p),
                ("pBitRevTable", _p), ("twidCoefModifier", C.c_uint16), ("bitRevFactor", C.c_uint16)]


class ArmFirQ15(C.Structure):     # arm_math.h:1027-1032
    _fields_ = [("numTaps", C.c_uint16), ("pState", _p), ("pCoeffs", _p)]


class Reference:
    """oracle/_ref/libmsdr_ref.so -- the reference's own sources compiled by oracle/build_ref.sh."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        L = self.lib
        L.arm_fir_init_q15.restype = C.c_int
        for n in ("calc_FIR_coeffs", "calc_FIR_coeffs_pid"):
            getattr(L, n).restype = None
            getattr(L, n).argtypes = [_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
        for n in ("Izero", "Izero_pid"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        for n in ("m_sinc", "m_sinc_pid"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_int, C.c_float]
        L.arm_sqrt_q31.argtypes = [C.c_int32, _p]

    def fir_q15_blocks(self, coeffs, x, block):
        coeffs = np.ascontiguousarray(coeffs, np.int16)
        x = np.ascontiguousarray(x, np.int16)
        # +8: the unrolled loop reads a few samples past the strict window
        st = np.full(coeffs.size + block + 8, 0x5A5A, np.int16)
        st[coeffs.size + block:] = 0
        S = ArmFirQ15()
        rc = self.lib.arm_fir_init_q15(C.byref(S), C.c_uint16(coeffs.size), _ptr(coeffs), _ptr(st),
                                       C.c_uint32(block))
        if rc != 0:
            return rc, None
        y = np.empty_like(x)
        for o in range(0, x.size, block):
            n = min(block, x.size - o)
            xin = x[o:o + n].copy()
            self.lib.arm_fir_fast_q15(C.byref(S), _ptr(xin), _ptr(y[o:o + n]), C.c_uint32(n))
        return 0, y

    def copy_q15(self, x):
        x = np.ascontiguousarray(x, np.int16)
        y = np.empty_like(x)
        self.lib.arm_copy_q15(_ptr(x), _ptr(y), C.c_uint32(x.size))
        return y

    def sqrt_q31(self, v):
        out = C.c_int32(0)
        rc = self.lib.arm_sqrt_q31(C.c_int32(int(v)), C.byref(out))
        return rc, out.value

    def calc_fir_coeffs(self, n, fc, astop=70.0, ftype=0, dfc=0.0, fs=24000.0, pi_double=False, room=None):
        buf = np.zeros(room or (2 * n + 8), np.int16)
        f = self.lib.calc_FIR_coeffs_pid if pi_double else self.lib.calc_FIR_coeffs
        f(_ptr(buf), int(n), float(fc), float(astop), int(ftype), float(dfc), float(fs))
        return buf

    # ---- row f4: the q15 FFT pieces (see oracle/build_ref.sh for what is and is not buildable) ----
    def table(self, name, n, ctype=C.c_int16):
        return np.array((ctype * n).in_dll(self.lib, name))

    def cfft64_q15(self, x):
        """arm_cfft_radix4_q15 (64 points, forward, bit reversal on): the reference's all-C entry to
        arm_radix4_butterfly_q15, in place on 64 {re, im} int16 pairs."""
        S = ArmCfftRadix4Q15()
        rc = self.lib.arm_cfft_radix4_init_q15(C.byref(S), C.c_uint16(64), C.c_uint8(0), C.c_uint8(1))
        assert rc == 0
        buf = np.ascontiguousarray(x, np.int16).copy()
        self.lib.arm_cfft_radix4_q15(C.byref(S), _ptr(buf))
        return buf

    def rfft128_q15(self, x):
        """What arm_rfft_q15(&FFT(128, 0, 1), x, out) computes, composed from the compiled stages (arm_rfft_q15
        itself needs the assembly-only arm_bitreversal_16): returns (out[256], x_after[128])."""
        if not hasattr(self, "_coefA"):
            self._coefA = self.table("realCoefAQ15", 8192)
            self._coefB = self.table("realCoefBQ15", 8192)
        work = self.cfft64_q15(x)
        out = np.zeros(256, np.int16)
        self.lib.arm_split_rfft_q15(_ptr(work), C.c_uint32(64), _ptr(self._coefA), _ptr(self._coefB), _ptr(out), C.c_uint32(64))
        return out, work
