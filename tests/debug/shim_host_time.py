"""tests/debug/shim_host_time.py -- `arm_fir_fast_q15(&FIR_I, I_buffer, I_FIR_out, AUDIO_BLOCK_SAMPLES)` on HOST arrays through msdr_cmsis_bind_host
(one channel: exactly the sketch's call, Minimal-SDR.ino:574-575): wall time per call, each call returns when I_FIR_out holds the result."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpuhelp import msdr  # noqa: E402
from test_gpu_cmsis import FirQ15, _sigs  # noqa: E402

ctx = msdr.Context(0)
lib = ctx.lib
_sigs(lib)
B = 128
rng = np.random.default_rng(2)
for ch in (1, 16):
    assert lib.msdr_cmsis_bind_host(ctx.h, ch) == 0
    FIR_I = FirQ15()
    taps = rng.integers(-2000, 2001, 102).astype(np.int16)
    st = np.zeros(512 + B, np.int16)
    assert lib.msdr_arm_fir_init_q15(C.byref(FIR_I), 102, taps.ctypes.data, st.ctypes.data, B) == 0
    src, dst = rng.integers(-20000, 20001, (ch, B)).astype(np.int16), np.empty((ch, B), np.int16)
    for _ in range(200):
        lib.msdr_arm_fir_fast_q15(C.byref(FIR_I), src.ctypes.data, dst.ctypes.data, B)
    K = 3000
    t0 = time.perf_counter()
    for _ in range(K):
        lib.msdr_arm_fir_fast_q15(C.byref(FIR_I), src.ctypes.data, dst.ctypes.data, B)
    us = (time.perf_counter() - t0) / K * 1e6
    print("host arrays, channels %2d: %.1f us per arm_fir_fast_q15 call  (MSDR_NO_BLOCK=%s)" % (ch, us, os.environ.get("MSDR_NO_BLOCK", "")), flush=True)
