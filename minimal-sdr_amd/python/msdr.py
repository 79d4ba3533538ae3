"""ctypes binding of the C ABI in include/msdr.h (minimal-sdr_amd/lib/libmsdr.so).

This is the stub a Python caller binds; the product is the shared library.  Used by tests/ and
bench.py.  It never falls back to a CPU implementation: if the library or a gfx950 device is
missing, construction raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libmsdr.so")

MODE_SYNCAM, MODE_AM, MODE_LSB, MODE_USB, MODE_CW = 0, 1, 2, 3, 4
SQRT_F32, SQRT_Q31 = 0, 1
MIXER_FS4, MIXER_NCO = 0, 1
ARITH_Q15, ARITH_F32 = 0, 1
BQ_LOWPASS, BQ_HIGHPASS, BQ_BANDPASS, BQ_NOTCH, BQ_LOWSHELF, BQ_HIGHSHELF = range(6)
AUDIO_SAMPLE_RATE_EXACT = 44117.64706
MAX_TAPSETS = 8
CHAIN_NO_TAP_FOLDING = 1
CHAIN_NO_MFMA = 8
CHAIN_SYNCAM_PLL = 32
CHAIN_FOLD_ANY_PERIOD = 64
CHAIN_OUT_I16 = 128
FE_DCBLOCK, FE_AMP, FE_AGC, FE_ALL = 1, 2, 4, 7

STATUS_ARGUMENT_ERROR, STATUS_LENGTH_ERROR, STATUS_NO_DEVICE = -1, -2, -100

_p = C.c_void_p


class MsdrError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("msdr status %d: %s" % (status, text))
        self.status = status


class ChainConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("arith", C.c_int32), ("channels", C.c_uint32), ("mixer", C.c_int32),
                ("num_taps", C.c_uint32), ("num_tapsets", C.c_uint32),
                ("coeffs_i", _p * MAX_TAPSETS), ("coeffs_q", _p * MAX_TAPSETS),
                ("default_mode", C.c_int32), ("mode", _p), ("tapset", _p), ("sqrt_kind", C.c_int32),
                ("osc_len", C.c_uint32), ("osc_i", _p), ("osc_q", _p),
                ("in_scale", C.c_float), ("num_biquad_stages", C.c_uint32), ("biquad_coeffs", _p),
                ("num_biquad_nodes", C.c_uint32), ("node_stages", C.c_uint32 * 2), ("node_coefs", _p * 2),
                ("time_segments", C.c_uint32), ("biquad_warmup", C.c_uint32), ("flags", C.c_uint32)]


class ChainInfo(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("grid", C.c_uint32), ("block", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("time_segments", C.c_uint32), ("warmup", C.c_uint32), ("tile", C.c_uint32),
                ("taps_padded", C.c_uint32), ("mfma_ksteps", C.c_uint32)]


_lib = None


def load_library(path=None):
    """Load libmsdr.so (no GPU needed to load it or to use the host-side designers)."""
    global _lib
    if _lib is None:
        path = path or os.environ.get("MSDR_LIB", LIB_PATH)
        if not os.path.exists(path):
            raise MsdrError(-100, "%s not built: run `make -C minimal-sdr_amd` (or __graft_entry__.build())" % path)
        _lib = C.CDLL(path)
        _lib.msdr_last_error.restype = C.c_char_p
        _lib.msdr_version.restype = C.c_char_p
        _lib.msdr_ctx_stream.restype = _p
        _lib.msdr_calc_FIR_coeffs.restype = None
        _lib.msdr_calc_FIR_coeffs.argtypes = [_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
        _lib.msdr_calc_FIR_coeffs_pid.restype = None
        _lib.msdr_calc_FIR_coeffs_pid.argtypes = [_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
        _lib.msdr_biquad_design.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_double, _p]
        _lib.msdr_biquad_df1_f32_cascade_info.argtypes = [C.c_uint8, _p, _p, _p, _p]
        _lib.msdr_biquad_df1_f32_state_to_cmsis.argtypes = [C.c_uint8, _p, _p, _p]
        _lib.msdr_biquad_df1_f32_state_from_cmsis.argtypes = [C.c_uint8, _p, _p, _p]
        _lib.msdr_biquad_df1_f32_set_coeffs.argtypes = [_p, _p]
        _lib.msdr_biquad_df1_f32_get_cmsis_state.argtypes = [_p, C.c_uint32, _p]
        _lib.msdr_fir_q15_set_coeffs.argtypes = [_p, _p]
        _lib.msdr_fir_f32_set_coeffs.argtypes = [_p, _p]
        _lib.msdr_chain_set_taps.argtypes = [_p, C.c_uint32, _p, _p]
        _lib.msdr_chain_set_osc.argtypes = [_p, _p, _p]
        _lib.msdr_chain_set_node_coefficients.argtypes = [_p, C.c_uint32, C.c_uint32, _p]
        _lib.msdr_chain_set_biquad_coeffs.argtypes = [_p, _p]
        _lib.msdr_malloc.argtypes = [_p, C.c_size_t, _p]
        _lib.msdr_free.argtypes = [_p, _p]
        _lib.msdr_memcpy_h2d.argtypes = [_p, _p, _p, C.c_size_t]
        _lib.msdr_memcpy_d2h.argtypes = [_p, _p, _p, C.c_size_t]
        _lib.msdr_memset.argtypes = [_p, _p, C.c_int, C.c_size_t]
        _lib.msdr_ctx_create.argtypes = [C.c_int, _p, _p]
        _lib.msdr_chain_process.argtypes = [_p, _p, _p, C.c_uint64]
        _lib.msdr_chain_graph_create.argtypes = [_p, C.c_uint32, _p, _p, C.c_uint64, _p]
        _lib.msdr_chain_graph_launch.argtypes = [_p]
        _lib.msdr_chain_graph_destroy.argtypes = [_p]
        _lib.msdr_chain_get_kernel_time.argtypes = [_p, _p, _p, C.c_int]
        for n in ("msdr_fir_q15_process", "msdr_fir_f32_process", "msdr_biquad_df1_f32_process"):
            getattr(_lib, n).argtypes = [_p, _p, _p, C.c_uint32]
        _lib.msdr_biquad_q15_update.argtypes = [_p, _p, C.c_uint32]
        _lib.msdr_frontend_update.argtypes = [_p, _p, _p, C.c_uint32, C.c_uint32]
        _lib.msdr_frontend_prime.argtypes = [_p, _p, C.c_uint32]
        _lib.msdr_frontend_get_state.argtypes = [_p, C.c_uint32, _p]
        _lib.msdr_amp_q15.argtypes = [_p, C.c_int32, _p, C.c_uint32, C.c_uint32, _p]
        _lib.msdr_anr_q15.argtypes = [_p, _p, C.c_int32, _p, C.c_uint32]
        _lib.msdr_anr_get_state.argtypes = [_p, C.c_uint32, _p]
        _lib.msdr_chain_set_anr.argtypes = [_p, _p, C.c_int32]
        _lib.msdr_dac_format_q15.argtypes = [_p, _p, _p, C.c_uint32, C.c_uint32]
        _lib.msdr_syncam_q15.argtypes = [_p, _p, _p, _p, _p, C.c_uint32]
        _lib.msdr_syncam_get_state.argtypes = [_p, C.c_uint32, _p]
        _lib.msdr_syncam_constants.argtypes = [_p]
        _lib.msdr_syncam_constants.restype = None
        _lib.msdr_fir_f32_set_input_range.argtypes = [_p, C.c_float]
        _lib.msdr_rfft128_tables.argtypes = [_p]
        _lib.msdr_rfft128_tables.restype = None
        _lib.msdr_rfft_q15_init_check.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.msdr_rfft128_q15.argtypes = [_p, _p, C.c_uint64, _p, _p, C.c_uint32]
        _lib.msdr_spectrum_create.argtypes = [_p, C.c_uint32, _p]
        _lib.msdr_spectrum_set_on.argtypes = [_p, C.c_int]
        _lib.msdr_spectrum_show.argtypes = [_p, _p, C.c_uint64, _p, _p, _p]
        _lib.msdr_spectrum_destroy.argtypes = [_p]
    return _lib


def _ck(rc):
    if rc != 0:
        raise MsdrError(rc, load_library().msdr_last_error().decode())


def _hp(a):
    return a.ctypes.data_as(_p) if a is not None else None


# ---- host-side designers (no device needed) ---------------------------------------------------
def calc_fir_coeffs(n, fc, astop=70.0, ftype=0, dfc=0.0, fs=24000.0, room=None, pi_double=False):
    """pi_double: `PI` as Arduino.h's double literal (what a Teensy build sees) instead of the vendored header's float."""
    buf = np.zeros(room or (2 * n + 8), np.int16)
    lib = load_library()
    f = lib.msdr_calc_FIR_coeffs_pid if pi_double else lib.msdr_calc_FIR_coeffs
    f(_hp(buf), int(n), float(fc), float(astop), int(ftype), float(dfc), float(fs))
    return buf


def biquad_design(kind, freq, q_or_gain, slope=1.0, fs=AUDIO_SAMPLE_RATE_EXACT):
    c = np.zeros(5, np.int32)
    _ck(load_library().msdr_biquad_design(int(kind), float(freq), float(q_or_gain), float(slope), float(fs), _hp(c)))
    return c


def biquad_cascade_info(coeffs):
    """(kappa, fp32_noise, cmsis_order) of a df1 cascade {b0,b1,b2,a1,a2} x stages -- host only."""
    c = np.ascontiguousarray(coeffs, np.float32).reshape(-1)
    k, nz, seq = C.c_double(0), C.c_double(0), C.c_int(0)
    _ck(load_library().msdr_biquad_df1_f32_cascade_info(C.c_uint8(c.size // 5), _hp(c) if c.size else None, C.byref(k), C.byref(nz), C.byref(seq)))
    return k.value, nz.value, bool(seq.value)


def biquad_state_to_cmsis(coeffs, lib_state):
    """The block-parallel kernels' 16-float state record -> arm_biquad_cascade_df1_f32's pState (4 per stage) -- host only."""
    c = np.ascontiguousarray(coeffs, np.float32).reshape(-1)
    st = np.ascontiguousarray(lib_state, np.float32).reshape(16)
    out = np.zeros(4 * (c.size // 5), np.float32)
    _ck(load_library().msdr_biquad_df1_f32_state_to_cmsis(C.c_uint8(c.size // 5), _hp(c), _hp(st), _hp(out)))
    return out


def biquad_state_from_cmsis(coeffs, pstate, d_hist=None):
    """pState -> the 16-float record (entries 0..7: the cascade's input history, taken from d_hist beyond the two newest)."""
    c = np.ascontiguousarray(coeffs, np.float32).reshape(-1)
    ps = np.ascontiguousarray(pstate, np.float32).reshape(-1)
    st = np.zeros(16, np.float32)
    if d_hist is not None:
        st[:len(d_hist)] = d_hist
    _ck(load_library().msdr_biquad_df1_f32_state_from_cmsis(C.c_uint8(c.size // 5), _hp(c), _hp(ps), _hp(st)))
    return st


class DeviceView:
    ptr = None


class DeviceArray:
    """A typed device buffer owned through msdr_malloc/msdr_free."""

    def __init__(self, ctx, shape, dtype):
        self.ctx, self.shape, self.dtype = ctx, tuple(np.atleast_1d(shape)), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        ptr = _p()
        _ck(ctx.lib.msdr_malloc(ctx.h, self.nbytes, C.byref(ptr)))
        self.ptr = ptr.value
        ctx._adopt(self)

    def upload(self, a):
        a = np.ascontiguousarray(a, self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        _ck(self.ctx.lib.msdr_memcpy_h2d(self.ctx.h, self.ptr, _hp(a), self.nbytes))
        return self

    def download(self):
        out = np.empty(self.shape, self.dtype)
        _ck(self.ctx.lib.msdr_memcpy_d2h(self.ctx.h, _hp(out), self.ptr, self.nbytes))
        return out

    def offset(self, nbytes):
        """A non-owning view starting `nbytes` into this buffer (pointer arithmetic only)."""
        v = DeviceView()
        v.ptr = self.ptr + int(nbytes)
        return v

    def fill(self, byte):
        _ck(self.ctx.lib.msdr_memset(self.ctx.h, self.ptr, int(byte), self.nbytes))
        return self

    def free(self):
        if self.ptr:
            if getattr(self.ctx, "h", None):
                self.ctx.lib.msdr_free(self.ctx.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = _p()
        _ck(self.lib.msdr_ctx_create(int(device), _p(stream) if stream else None, C.byref(h)))
        self.h = h

    def synchronize(self):
        _ck(self.lib.msdr_ctx_synchronize(self.h))

    def stream(self):
        """The hipStream_t the context enqueues on, as an int (0 = the NULL stream); e.g. msdr_dist.OverlappedGather(compute_stream=...)."""
        return int(self.lib.msdr_ctx_stream(self.h) or 0)

    def array(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def to_device(self, a):
        a = np.ascontiguousarray(a)
        return DeviceArray(self, a.shape, a.dtype).upload(a)

    # ---- stateless stages -----------------------------------------------------------------------
    def enable_kernel_timing(self, on=True):
        """HIP events around the main kernel of the FIR stage mirrors (msdr_ctx_enable_kernel_timing)."""
        _ck(self.lib.msdr_ctx_enable_kernel_timing(self.h, int(on)))

    def kernel_time(self, reset=True):
        ms, n = C.c_double(0), C.c_uint64(0)
        _ck(self.lib.msdr_ctx_get_kernel_time(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def mix_fs4_q15(self, d_x, d_i, d_q, channels, n):
        _ck(self.lib.msdr_mix_fs4_q15(self.h, _p(d_x.ptr), _p(d_i.ptr), _p(d_q.ptr), C.c_uint32(channels), C.c_uint32(n)))

    def freqconv_q15(self, d_i, d_q, osc_i, osc_q, direction, passthrough, channels, n):
        oi, oq = np.ascontiguousarray(osc_i, np.int16), np.ascontiguousarray(osc_q, np.int16)
        _ck(self.lib.msdr_freqconv_q15(self.h, _p(d_i.ptr), _p(d_q.ptr), _hp(oi), _hp(oq), C.c_uint32(oi.size),
                                       int(direction), int(passthrough), C.c_uint32(channels), C.c_uint32(n)))

    def freqconv_f32(self, d_i, d_q, osc_i, osc_q, direction, passthrough, channels, n):
        oi, oq = np.ascontiguousarray(osc_i, np.float32), np.ascontiguousarray(osc_q, np.float32)
        _ck(self.lib.msdr_freqconv_f32(self.h, _p(d_i.ptr), _p(d_q.ptr), _hp(oi), _hp(oq), C.c_uint32(oi.size),
                                       int(direction), int(passthrough), C.c_uint32(channels), C.c_uint32(n)))

    def demod_q15(self, mode, d_i, d_q, d_out, channels, n, sqrt_kind=SQRT_F32, d_mode=None):
        _ck(self.lib.msdr_demod_q15(self.h, int(mode), _p(d_mode.ptr) if d_mode else None, int(sqrt_kind),
                                    _p(d_i.ptr), _p(d_q.ptr), _p(d_out.ptr), C.c_uint32(channels), C.c_uint32(n)))

    def demod_f32(self, mode, d_i, d_q, d_out, channels, n, d_mode=None):
        _ck(self.lib.msdr_demod_f32(self.h, int(mode), _p(d_mode.ptr) if d_mode else None,
                                    _p(d_i.ptr), _p(d_q.ptr), _p(d_out.ptr), C.c_uint32(channels), C.c_uint32(n)))

    def close(self):
        """Destroys the context.  Objects created on it that are still alive are released first: a handle must never reach
        the library after its context is gone (the garbage collector may run an object's __del__ long after the fixture that
        owned the context has closed it)."""
        if self.h:
            for obj in list(getattr(self, "_children", ())):
                try:
                    obj.close() if hasattr(obj, "close") else obj.free()
                except Exception:
                    pass
            self.lib.msdr_ctx_destroy(self.h)
            self.h = None

    def _adopt(self, obj):
        if not hasattr(self, "_children"):
            import weakref
            self._children = weakref.WeakSet()
        self._children.add(obj)


class _Instance:
    _destroy = None

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name == "ctx" and value is not None and hasattr(value, "_adopt"):
            value._adopt(self)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):            # context already destroyed: its objects went with it
                getattr(self.ctx.lib, self._destroy)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FirQ15(_Instance):
    """arm_fir_init_q15 / arm_fir_fast_q15, batched over channels."""
    _destroy = "msdr_fir_q15_destroy"

    def __init__(self, ctx, coeffs, channels):
        self.ctx = ctx
        c = np.ascontiguousarray(coeffs, np.int16)
        self.ntaps = int(c.size)
        h = _p()
        _ck(ctx.lib.msdr_fir_q15_create(ctx.h, C.c_uint16(c.size), _hp(c), C.c_uint32(channels), C.byref(h)))
        self.h = h

    def process(self, d_src, d_dst, n):
        _ck(self.ctx.lib.msdr_fir_q15_process(self.h, d_src.ptr, d_dst.ptr, n))

    def reset(self):
        _ck(self.ctx.lib.msdr_fir_q15_reset(self.h))

    def set_coeffs(self, coeffs):
        """pCoeffs rewritten under the running filter (UI.cpp:337-345): state kept."""
        c = np.ascontiguousarray(coeffs, np.int16)
        if c.size != self.ntaps:
            raise ValueError("set_coeffs: %d taps given, the instance has %d" % (c.size, self.ntaps))
        _ck(self.ctx.lib.msdr_fir_q15_set_coeffs(self.h, _hp(c)))


class FirF32(_Instance):
    """arm_fir_init_f32 / arm_fir_f32, batched over channels."""
    _destroy = "msdr_fir_f32_destroy"

    def __init__(self, ctx, coeffs, channels):
        self.ctx = ctx
        c = np.ascontiguousarray(coeffs, np.float32)
        self.ntaps = int(c.size)
        h = _p()
        _ck(ctx.lib.msdr_fir_f32_create(ctx.h, C.c_uint16(c.size), _hp(c), C.c_uint32(channels), C.byref(h)))
        self.h = h

    def process(self, d_src, d_dst, n):
        _ck(self.ctx.lib.msdr_fir_f32_process(self.h, d_src.ptr, d_dst.ptr, n))

    def reset(self):
        _ck(self.ctx.lib.msdr_fir_f32_reset(self.h))

    def set_coeffs(self, coeffs):
        c = np.ascontiguousarray(coeffs, np.float32)
        if c.size != self.ntaps:
            raise ValueError("set_coeffs: %d taps given, the instance has %d" % (c.size, self.ntaps))
        _ck(self.ctx.lib.msdr_fir_f32_set_coeffs(self.h, _hp(c)))

    def set_input_range(self, max_abs):
        _ck(self.ctx.lib.msdr_fir_f32_set_input_range(self.h, C.c_float(max_abs)))

    def kernel_name(self):
        f = self.ctx.lib.msdr_fir_f32_kernel_name
        f.restype = C.c_char_p
        f.argtypes = [_p]
        return f(self.h).decode()


class BiquadDf1F32(_Instance):
    """arm_biquad_cascade_df1_init_f32 / arm_biquad_cascade_df1_f32, batched over channels."""
    _destroy = "msdr_biquad_df1_f32_destroy"

    def __init__(self, ctx, coeffs, channels):
        self.ctx = ctx
        c = np.ascontiguousarray(coeffs, np.float32).reshape(-1)
        self.stages = int(c.size // 5)
        h = _p()
        _ck(ctx.lib.msdr_biquad_df1_f32_create(ctx.h, C.c_uint8(c.size // 5), _hp(c) if c.size else None,
                                               C.c_uint32(channels), C.byref(h)))
        self.h = h

    def process(self, d_src, d_dst, n):
        _ck(self.ctx.lib.msdr_biquad_df1_f32_process(self.h, d_src.ptr, d_dst.ptr, n))

    def reset(self):
        _ck(self.ctx.lib.msdr_biquad_df1_f32_reset(self.h))

    def set_coeffs(self, coeffs):
        """pCoeffs rewritten under the running cascade: continues from CMSIS' pState."""
        c = np.ascontiguousarray(coeffs, np.float32).reshape(-1)
        if c.size != 5 * self.stages:
            raise ValueError("set_coeffs: %d values given, the cascade has %d stages x 5" % (c.size, self.stages))
        _ck(self.ctx.lib.msdr_biquad_df1_f32_set_coeffs(self.h, _hp(c)))

    def cmsis_state(self, channel, stages):
        out = np.zeros(4 * stages, np.float32)
        _ck(self.ctx.lib.msdr_biquad_df1_f32_get_cmsis_state(self.h, C.c_uint32(channel), _hp(out)))
        return out


class BiquadQ15(_Instance):
    """AudioFilterBiquad: setCoefficients(stage, coef[5]) / update(), batched over channels."""
    _destroy = "msdr_biquad_q15_destroy"

    def __init__(self, ctx, channels):
        self.ctx = ctx
        h = _p()
        _ck(ctx.lib.msdr_biquad_q15_create(ctx.h, C.c_uint32(channels), C.byref(h)))
        self.h = h

    def set_coefficients(self, stage, coef):
        c = np.ascontiguousarray(coef, np.int32)
        _ck(self.ctx.lib.msdr_biquad_q15_set_coefficients(self.h, C.c_uint32(stage), _hp(c)))

    def update(self, d_data, n):
        _ck(self.ctx.lib.msdr_biquad_q15_update(self.h, d_data.ptr, n))

    def definition(self, channel=0):
        d = np.zeros(32, np.int32)
        _ck(self.ctx.lib.msdr_biquad_q15_get_definition(self.h, C.c_uint32(channel), _hp(d)))
        return d


class Frontend(_Instance):
    """adc1's DC block + amp_adc + AGC(), batched over channels; state per channel as the sketch's statics."""
    _destroy = "msdr_frontend_destroy"

    def __init__(self, ctx, channels):
        self.ctx = ctx
        h = _p()
        _ck(ctx.lib.msdr_frontend_create(ctx.h, C.c_uint32(channels), C.byref(h)))
        self.h = h

    def prime(self, first_conversion):
        v = np.ascontiguousarray(np.atleast_1d(first_conversion), np.uint16)
        _ck(self.ctx.lib.msdr_frontend_prime(self.h, _hp(v), C.c_uint32(v.size)))

    def set_agc(self, on):
        _ck(self.ctx.lib.msdr_frontend_set_agc(self.h, int(bool(on))))

    def gain(self, n):
        _ck(self.ctx.lib.msdr_frontend_gain(self.h, C.c_float(n)))

    def update(self, d_adc, d_out, n, stages=FE_ALL):
        _ck(self.ctx.lib.msdr_frontend_update(self.h, d_adc.ptr if hasattr(d_adc, "ptr") else d_adc,
                                              d_out.ptr if hasattr(d_out, "ptr") else d_out, C.c_uint32(n), C.c_uint32(stages)))

    def state(self, channel=0):
        st = np.zeros(32, np.int32)
        _ck(self.ctx.lib.msdr_frontend_get_state(self.h, C.c_uint32(channel), _hp(st)))
        return st


class Syncam(_Instance):
    """The PLL branch of the demod switch (Minimal-SDR.ino:631-688), batched over channels."""
    _destroy = "msdr_syncam_destroy"

    def __init__(self, ctx, channels):
        self.ctx = ctx
        h = _p()
        _ck(ctx.lib.msdr_syncam_create(ctx.h, C.c_uint32(channels), C.byref(h)))
        self.h = h

    def process(self, d_i, d_q, d_out, n, d_mode=None):
        _ck(self.ctx.lib.msdr_syncam_q15(self.h, d_mode.ptr if d_mode is not None else None, d_i.ptr, d_q.ptr, d_out.ptr, C.c_uint32(n)))

    def reset(self):
        _ck(self.ctx.lib.msdr_syncam_reset(self.h))

    def state(self, channel=0):
        st = np.zeros(3, np.float32)
        _ck(self.ctx.lib.msdr_syncam_get_state(self.h, C.c_uint32(channel), _hp(st)))
        return st


class Anr(_Instance):
    """LMS automatic notch / noise reduction (Minimal-SDR.ino:702-770), batched over channels."""
    _destroy = "msdr_anr_destroy"

    def __init__(self, ctx, channels):
        self.ctx = ctx
        h = _p()
        _ck(ctx.lib.msdr_anr_create(ctx.h, C.c_uint32(channels), C.byref(h)))
        self.h = h

    def process(self, d_data, n, anr_on=1, d_anr_on=None):
        _ck(self.ctx.lib.msdr_anr_q15(self.h, d_anr_on.ptr if d_anr_on is not None else None, C.c_int32(anr_on), d_data.ptr, C.c_uint32(n)))

    def reset(self):
        _ck(self.ctx.lib.msdr_anr_reset(self.h))

    def state(self, channel=0):
        st = np.zeros(196, np.float32)
        _ck(self.ctx.lib.msdr_anr_get_state(self.h, C.c_uint32(channel), _hp(st)))
        return st


def dac_format_q15(ctx, d_src, d_dest, channels, n):
    _ck(ctx.lib.msdr_dac_format_q15(ctx.h, d_src.ptr if d_src is not None else None, d_dest.ptr, C.c_uint32(channels), C.c_uint32(n)))


def rfft128_tables():
    """(twiddleCoef_64_q15[96], realCoefAQ15[::64 pairs][128], realCoefBQ15[::64 pairs][128]) -- host side."""
    t = np.zeros(352, np.int16)
    load_library().msdr_rfft128_tables(_hp(t))
    return t[:96], t[96:224], t[224:]


def rfft128_q15(ctx, d_src, src_stride, nfft, d_fft_out=None, d_columns=None):
    """arm_rfft_q15(&FFT(128, 0, 1), ...) batched (UI.cpp:551) + showSpectrum's column heights (UI.cpp:557-572)."""
    _ck(ctx.lib.msdr_rfft128_q15(ctx.h, d_src.ptr, C.c_uint64(src_stride), d_fft_out.ptr if d_fft_out is not None else None,
                                 d_columns.ptr if d_columns is not None else None, C.c_uint32(nfft)))


class Spectrum(_Instance):
    """initSpectrum() / showSpectrum() (UI.cpp:520-592): Spectrum_on + the every-25th-call cadence."""
    _destroy = "msdr_spectrum_destroy"

    def __init__(self, ctx, channels):
        self.ctx = ctx
        h = _p()
        _ck(ctx.lib.msdr_spectrum_create(ctx.h, C.c_uint32(channels), C.byref(h)))
        self.h = h

    def set_on(self, on):
        _ck(self.ctx.lib.msdr_spectrum_set_on(self.h, int(on)))

    def show(self, d_data, channel_stride, d_fft_out=None, d_columns=None):
        drawn = C.c_int(0)
        _ck(self.ctx.lib.msdr_spectrum_show(self.h, d_data.ptr, C.c_uint64(channel_stride),
                                            d_fft_out.ptr if d_fft_out is not None else None,
                                            d_columns.ptr if d_columns is not None else None, C.byref(drawn)))
        return bool(drawn.value)


def syncam_constants():
    c = np.zeros(4, np.float32)
    load_library().msdr_syncam_constants(_hp(c))
    return c


def amp_multiplier(n):
    lib = load_library()
    lib.msdr_amp_multiplier.restype = C.c_int32
    lib.msdr_amp_multiplier.argtypes = [C.c_float]
    return int(lib.msdr_amp_multiplier(C.c_float(n)))


def amp_q15(ctx, multiplier, d_data, channels, n):
    t = C.c_int(0)
    _ck(ctx.lib.msdr_amp_q15(ctx.h, C.c_int32(multiplier), d_data.ptr, C.c_uint32(channels), C.c_uint32(n), C.byref(t)))
    return bool(t.value)


class ChainGraph:
    """`ticks` consecutive block-cadence calls of one chain as ONE HIP graph (include/msdr.h: msdr_chain_graph_*)."""

    def __init__(self, chain, d_ifs, d_audios, n):
        assert len(d_ifs) == len(d_audios)
        self.chain, self.ticks = chain, len(d_ifs)
        pi = (C.c_void_p * self.ticks)(*[(d.ptr if hasattr(d, "ptr") else int(d)) for d in d_ifs])
        po = (C.c_void_p * self.ticks)(*[(d.ptr if hasattr(d, "ptr") else int(d)) for d in d_audios])
        self.h = C.c_void_p()
        _ck(chain.ctx.lib.msdr_chain_graph_create(chain.h, self.ticks, pi, po, n, C.byref(self.h)))

    def launch(self):
        _ck(self.chain.ctx.lib.msdr_chain_graph_launch(self.h))

    def close(self):
        if self.h:
            self.chain.ctx.lib.msdr_chain_graph_destroy(self.h)
            self.h = None


class Chain(_Instance):
    """The fused chain (msdr_chain_*): demodulation() + the biquad nodes behind it."""
    _destroy = "msdr_chain_destroy"

    def __init__(self, ctx, arith, channels, coeffs_i, coeffs_q, mixer=MIXER_FS4, mode=MODE_AM, modes=None,
                 tapsets=None, osc_i=None, osc_q=None, sqrt_kind=SQRT_F32, in_scale=0.0, biquad_coeffs=None,
                 biquad_nodes=(), time_segments=0, biquad_warmup=0, flags=0):
        self.ctx, self.arith, self.channels = ctx, arith, channels
        tdt = np.float32 if arith == ARITH_F32 else np.int16
        ci = [np.ascontiguousarray(c, tdt) for c in (coeffs_i if isinstance(coeffs_i, (list, tuple)) else [coeffs_i])]
        cq = [np.ascontiguousarray(c, tdt) for c in (coeffs_q if isinstance(coeffs_q, (list, tuple)) else [coeffs_q])]
        keep = [ci, cq]
        cfg = ChainConfig()
        cfg.struct_size = C.sizeof(ChainConfig)
        cfg.arith, cfg.channels, cfg.mixer = arith, channels, mixer
        cfg.num_taps, cfg.num_tapsets = ci[0].size, len(ci)
        self.ntaps, self.osc_len, self.stages = int(ci[0].size), 0, 0     # what the setters' arrays must hold (the C side copies that many elements)
        for k in range(len(ci)):
            cfg.coeffs_i[k], cfg.coeffs_q[k] = _hp(ci[k]), _hp(cq[k])
        cfg.default_mode = mode
        if modes is not None:
            m = np.ascontiguousarray(modes, np.int32)
            keep.append(m)
            cfg.mode = _hp(m)
        if tapsets is not None:
            t = np.ascontiguousarray(tapsets, np.int32)
            keep.append(t)
            cfg.tapset = _hp(t)
        cfg.sqrt_kind = sqrt_kind
        if osc_i is not None:
            oi, oq = np.ascontiguousarray(osc_i, tdt), np.ascontiguousarray(osc_q, tdt)
            keep += [oi, oq]
            cfg.osc_len, cfg.osc_i, cfg.osc_q = oi.size, _hp(oi), _hp(oq)
            self.osc_len = int(oi.size)
        cfg.in_scale = in_scale
        if biquad_coeffs is not None and arith == ARITH_F32:
            b = np.ascontiguousarray(biquad_coeffs, np.float32).reshape(-1)
            keep.append(b)
            cfg.num_biquad_stages, cfg.biquad_coeffs = b.size // 5, _hp(b)
            self.stages = int(b.size // 5)
        if arith == ARITH_Q15:
            cfg.num_biquad_nodes = len(biquad_nodes)
            for k, stages in enumerate(biquad_nodes):
                s = np.ascontiguousarray(stages, np.int32).reshape(-1)
                keep.append(s)
                cfg.node_stages[k], cfg.node_coefs[k] = s.size // 5, _hp(s)
        cfg.time_segments, cfg.biquad_warmup, cfg.flags = time_segments, biquad_warmup, flags
        h = _p()
        _ck(ctx.lib.msdr_chain_create(ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h

    def process(self, d_if, d_audio, n):
        """d_if / d_audio: DeviceArray or raw device pointers (int)."""
        pi = d_if.ptr if hasattr(d_if, "ptr") else int(d_if)
        po = d_audio.ptr if hasattr(d_audio, "ptr") else int(d_audio)
        _ck(self.ctx.lib.msdr_chain_process(self.h, pi, po, n))

    def graph(self, d_ifs, d_audios, n):
        """msdr_chain_graph_create: len(d_ifs) (even) consecutive calls of n samples as one HIP graph; .launch() enqueues them."""
        return ChainGraph(self, d_ifs, d_audios, n)

    def reset(self):
        _ck(self.ctx.lib.msdr_chain_reset(self.h))

    def init_fir(self):
        """init_FIR() (Minimal-SDR.ino:901-930): FIR state only."""
        _ck(self.ctx.lib.msdr_chain_init_fir(self.h))

    def set_mode(self, channel, mode, tapset=0):
        _ck(self.ctx.lib.msdr_chain_set_mode(self.h, C.c_uint32(channel), C.c_int32(mode), C.c_int32(tapset)))

    def set_anr(self, anr_on=None, anr_on_all=0):
        a = np.ascontiguousarray(anr_on, np.int32) if anr_on is not None else None
        _ck(self.ctx.lib.msdr_chain_set_anr(self.h, _hp(a), C.c_int32(anr_on_all)))

    # ---- live updates, every filter state kept (include/msdr.h) ----
    def set_taps(self, tapset, coeffs_i, coeffs_q):
        tdt = np.float32 if self.arith == ARITH_F32 else np.int16
        ci, cq = np.ascontiguousarray(coeffs_i, tdt), np.ascontiguousarray(coeffs_q, tdt)
        if ci.size != self.ntaps or cq.size != self.ntaps:
            raise ValueError("set_taps: %d / %d taps given, the chain was created with %d (numTaps is fixed at creation, as in CMSIS)" % (ci.size, cq.size, self.ntaps))
        _ck(self.ctx.lib.msdr_chain_set_taps(self.h, C.c_uint32(tapset), _hp(ci), _hp(cq)))

    def set_osc(self, osc_i, osc_q):
        tdt = np.float32 if self.arith == ARITH_F32 else np.int16
        oi, oq = np.ascontiguousarray(osc_i, tdt), np.ascontiguousarray(osc_q, tdt)
        if self.osc_len and (oi.size != self.osc_len or oq.size != self.osc_len):
            raise ValueError("set_osc: tables of %d / %d entries, the chain's have %d" % (oi.size, oq.size, self.osc_len))
        _ck(self.ctx.lib.msdr_chain_set_osc(self.h, _hp(oi), _hp(oq)))

    def set_node_coefficients(self, node, stage, coef):
        c = np.ascontiguousarray(coef, np.int32)
        if c.size != 5:
            raise ValueError("set_node_coefficients: one stage = 5 words (b0, b1, b2, a1, a2), %d given" % c.size)
        _ck(self.ctx.lib.msdr_chain_set_node_coefficients(self.h, C.c_uint32(node), C.c_uint32(stage), _hp(c)))

    def set_biquad_coeffs(self, coeffs):
        c = np.ascontiguousarray(coeffs, np.float32).reshape(-1)
        if self.stages and c.size != 5 * self.stages:
            raise ValueError("set_biquad_coeffs: %d values given, the chain's cascade has %d stages x 5" % (c.size, self.stages))
        _ck(self.ctx.lib.msdr_chain_set_biquad_coeffs(self.h, _hp(c)))

    def info(self):
        i = ChainInfo()
        _ck(self.ctx.lib.msdr_chain_get_info(self.h, C.byref(i)))
        return {"kernel": i.kernel.decode(), "grid": i.grid, "block": i.block, "lds_bytes": i.lds_bytes,
                "time_segments": i.time_segments, "warmup": i.warmup, "tile": i.tile, "taps_padded": i.taps_padded,
                "mfma_ksteps": i.mfma_ksteps}

    def enable_timing(self, on=True):
        _ck(self.ctx.lib.msdr_chain_enable_timing(self.h, int(on)))

    def kernel_time(self, reset=True):
        ms, n = C.c_double(0), C.c_uint64(0)
        _ck(self.ctx.lib.msdr_chain_get_kernel_time(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value


GATHER_SLOTS = 4


def comm_unique_id():
    """128 opaque bytes made by rank 0 (ncclGetUniqueId); pass them to the other ranks by any host channel."""
    buf = C.create_string_buffer(128)
    _ck(load_library().msdr_comm_get_unique_id(buf))
    return buf.raw


class Comm:
    """msdr_comm_*: the RCCL gather of demodulated audio, driven from C (include/msdr.h)."""

    def __init__(self, ctx, unique_id, rank, world):
        self.ctx, self.rank, self.world = ctx, rank, world
        h = _p()
        _ck(ctx.lib.msdr_comm_create(ctx.h, C.c_char_p(unique_id), int(rank), int(world), C.byref(h)))
        self.h = h

    def begin(self, slot, d_local_ptr, local_bytes, d_recv_ptr, root):
        _ck(self.ctx.lib.msdr_gather_audio_begin(self.h, int(slot), _p(d_local_ptr), C.c_size_t(local_bytes),
                                                 _p(d_recv_ptr) if d_recv_ptr else None, int(root)))

    def wait(self, slot, host_wait=False):
        _ck(self.ctx.lib.msdr_gather_audio_wait(self.h, int(slot), int(host_wait)))

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                self.ctx.lib.msdr_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
