#!/usr/bin/env python3
"""Generate tests/golden/golden.npz + manifest.json.  Runs ONLY in the build container (needs
/root/reference and oracle/_ref/libmsdr_ref.so = the reference's own sources compiled by
oracle/build_ref.sh).  The committed fixtures are numbers, not code:

 F1  tap sets: the four static 86-tap tables (Minimal-SDR.ino:119-128, parsed as data) and
     calc_FIR_coeffs outputs from the COMPILED REFERENCE designer (.ino:782-899).
 F2  FIR known answers: outputs of the COMPILED REFERENCE arm_fir_fast_q15 (+ init/copy/sqrt_q31).
 F3  chain known answers: Fs/4 mix and demod switch restated in numpy in THIS file (independent
     of oracle/msdr_oracle.c) around the COMPILED REFERENCE FIR -> I/Q intermediates and audio.
Nothing here imports the C oracle; the oracle is checked AGAINST these vectors.
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orclib  # noqa: E402

REF_ROOT = os.environ.get("MSDR_REFERENCE", "/root/reference")
FS = 24000.0
B = 128


def parse_tables():
    src = open(os.path.join(REF_ROOT, "Minimal-SDR.ino")).read()
    out = {}
    for name in ("FIR_SSB_I_coeffs", "FIR_SSB_Q_coeffs", "FIR_CW_I_coeffs", "FIR_CW_Q_coeffs"):
        m = re.search(r"const int16_t %s\[[^\]]*\]\s*=\s*\{([^}]*)\}" % name, src)
        out[name] = np.array([int(v) for v in m.group(1).split(",")], np.int16)
        assert out[name].size == 86
    return out


def signals():
    n = np.arange(16 * B)
    rng = np.random.default_rng(1)
    am = np.round(12000 * (0.5 + 0.4 * np.sin(2 * np.pi * 300 * n / FS)) * np.cos(2 * np.pi * 6000 * n / FS))
    am = (am + rng.integers(-200, 201, n.size)).astype(np.int16)
    tones = np.round(6000 * np.cos(2 * np.pi * 6700 * n / FS) + 6000 * np.cos(2 * np.pi * 5300 * n / FS + 0.3))
    tones = tones.astype(np.int16)
    noise = np.random.default_rng(3).integers(-8000, 8001, n.size).astype(np.int16)
    full = np.random.default_rng(4).integers(-32768, 32768, n.size).astype(np.int16)
    full[::17] = -32768
    full[5::23] = 32767
    return {"am": am, "tones": tones, "noise": noise, "full": full}


def np_mix_fs4(x):
    """Minimal-SDR.ino:546-558 in numpy (int16 wrap on negate)."""
    x = x.astype(np.int32)
    k = np.arange(x.size) & 3
    neg = ((-x + 32768) % 65536 - 32768)
    i = np.where(k == 0, x, np.where(k == 2, neg, 0)).astype(np.int16)
    q = np.where(k == 1, x, np.where(k == 3, neg, 0)).astype(np.int16)
    return i, q


def wrap16(v):
    return ((np.asarray(v, np.int64) + 32768) % 65536 - 32768).astype(np.int16)


def np_demod(mode, i, q, ref, sqrt_kind):
    """Minimal-SDR.ino:589-627 in numpy."""
    i32, q32 = i.astype(np.int64), q.astype(np.int64)
    if mode == orclib.LSB:
        return wrap16(i32 - q32)
    if mode == orclib.USB:
        return wrap16(i32 + q32)
    s = ((i32 * i32 + q32 * q32 + 2 ** 31) % 2 ** 32 - 2 ** 31)        # int32 wrap
    if sqrt_kind == orclib.SQRT_Q31:
        return np.array([ref.sqrt_q31(int(v))[1] >> 16 for v in s], np.int64).astype(np.int16)
    f = s.astype(np.float32)
    r = np.where(f >= 0, np.sqrt(np.maximum(f, 0), dtype=np.float32), np.float32(0))
    return wrap16(r.astype(np.int64))


def main():
    ref = orclib.Reference()
    g = {}
    meta = {"generator": "tests/golden/make_golden.py", "reference": "FrankBoesing/Minimal-SDR @ /root/reference",
            "entries": {}}

    # ---- F1 tap sets ---------------------------------------------------------------------
    for k, v in parse_tables().items():
        g["taps/" + k] = v
    designs = [(102, 2800, 70, 0, 0), (62, 2800, 70, 0, 0), (256, 2800, 70, 0, 0), (512, 2800, 70, 0, 0),
               (102, 600, 70, 0, 0), (102, 5000, 70, 0, 0), (102, 2800, 40, 0, 0), (102, 2800, 15, 0, 0),
               (64, 3000, 60, 1, 0), (64, 6000, 60, 2, 1500), (64, 6000, 60, 3, 800), (32, 0, 70, 4, 0),
               (100, 1330, 70, 0, 0)]
    for (n, fc, a, t, dfc) in designs:
        for pid in (False, True):
            c = ref.calc_fir_coeffs(n, fc, a, t, dfc, FS, pi_double=pid)
            g["design/n%d_fc%d_a%d_t%d_d%d_%s" % (n, fc, a, t, dfc, "pid" if pid else "pif")] = c
    meta["design_cases"] = designs

    # ---- F2 FIR known answers (compiled reference arm_fir_fast_q15) -----------------------
    sig = signals()
    taps = {"ssb_i": g["taps/FIR_SSB_I_coeffs"], "ssb_q": g["taps/FIR_SSB_Q_coeffs"],
            "am102": g["design/n102_fc2800_a70_t0_d0_pif"][:102],
            "lp256": g["design/n256_fc2800_a70_t0_d0_pif"][:256],
            "lp512": g["design/n512_fc2800_a70_t0_d0_pif"][:512],
            "lp62": g["design/n62_fc2800_a70_t0_d0_pif"][:62],
            "wrap8": np.full(8, 32767, np.int16),            # forces the 32-bit accumulator to wrap
            "n4": np.array([1000, -2000, 3000, -4000], np.int16),
            "n6": np.array([30000, -30000, 30000, -30000, 30000, -30000], np.int16)}
    for tn, tv in taps.items():
        g["fir/taps_" + tn] = tv
        for sn in ("noise", "full"):
            x = sig[sn][:8 * B]
            for blk in (128, 130, 7):
                rc, y = ref.fir_q15_blocks(tv, x, blk)
                assert rc == 0
                g["fir/%s_%s_b%d" % (tn, sn, blk)] = y
    g["fir/x_noise"] = sig["noise"][:8 * B]
    g["fir/x_full"] = sig["full"][:8 * B]
    rc, _ = ref.fir_q15_blocks(np.ones(5, np.int16), sig["noise"][:B], B)
    meta["fir_init_odd_taps_status"] = int(rc)                 # arm_fir_init_q15.c:93-96
    sq_in = np.concatenate([np.array([0, 1, 2, 3, 4, 100, 32767, 65536, 2 ** 30, 2 ** 31 - 1, -1, -2 ** 31]),
                            np.random.default_rng(7).integers(1, 2 ** 31, 500)]).astype(np.int64)
    g["sqrt_q31/in"] = sq_in.astype(np.int32)
    g["sqrt_q31/out"] = np.array([ref.sqrt_q31(int(v))[1] for v in sq_in], np.int32)
    g["copy_q15/out"] = ref.copy_q15(sig["full"][:131])

    # ---- F3 chain known answers: numpy mix + reference FIR + numpy demod -----------------
    modes = {"AM": (orclib.AM, "am102", "am102"), "LSB": (orclib.LSB, "ssb_i", "ssb_q"),
             "USB": (orclib.USB, "ssb_i", "ssb_q"),
             "CW": (orclib.CW, g["taps/FIR_CW_I_coeffs"], g["taps/FIR_CW_Q_coeffs"])}
    for sn, x in sig.items():
        g["chain/x_" + sn] = x
        i, q = np_mix_fs4(x)
        for mn, (mode, ti, tq) in modes.items():
            ti = taps[ti] if isinstance(ti, str) else ti
            tq = taps[tq] if isinstance(tq, str) else tq
            _, fi = ref.fir_q15_blocks(ti, i, B)
            _, fq = ref.fir_q15_blocks(tq, q, B)
            g["chain/%s_%s_I" % (sn, mn)] = fi
            g["chain/%s_%s_Q" % (sn, mn)] = fq
            g["chain/%s_%s_audio" % (sn, mn)] = np_demod(mode, fi, fq, ref, orclib.SQRT_F32)
            if mn in ("AM", "CW"):
                g["chain/%s_%s_audio_q31" % (sn, mn)] = np_demod(mode, fi, fq, ref, orclib.SQRT_Q31)

    # ---- F6 spectrum FFT (row f4): reference tables as data + known answers of the compiled stages ----
    g["fft/twiddleCoef_64_q15"] = ref.table("twiddleCoef_64_q15", 96)
    g["fft/armBitRevIndexTable_fixed_64"] = ref.table("armBitRevIndexTable_fixed_64", 56, orclib.C.c_uint16)
    g["fft/realCoefAQ15_stride64"] = ref.table("realCoefAQ15", 8192).reshape(-1, 2)[::64].reshape(-1).copy()
    g["fft/realCoefBQ15_stride64"] = ref.table("realCoefBQ15", 8192).reshape(-1, 2)[::64].reshape(-1).copy()
    rng = np.random.default_rng(64)
    n128 = np.arange(128)
    fx = [sig["am"][:128], sig["tones"][128:256], sig["noise"][:128], sig["full"][:128],
          np.full(128, -32768, np.int16), np.full(128, 32767, np.int16),
          (8000 * np.sin(2 * np.pi * 10 * n128 / 128)).astype(np.int16),
          (30000 * np.cos(2 * np.pi * 32 * n128 / 128)).astype(np.int16),
          np.where(n128 & 1, -32768, 32767).astype(np.int16), np.zeros(128, np.int16)]
    fx += [rng.integers(-a, a + 1, 128).astype(np.int16) for a in (3, 200, 5000, 32767) for _ in range(4)]
    fx = np.stack(fx)
    g["fft/x"] = fx
    g["fft/rfft128_out"] = np.stack([ref.rfft128_q15(x)[0] for x in fx])
    g["fft/rfft128_work"] = np.stack([ref.rfft128_q15(x)[1] for x in fx])

    path = os.path.join(HERE, "golden.npz")
    np.savez_compressed(path, **g)
    for k, v in g.items():
        meta["entries"][k] = {"dtype": str(v.dtype), "shape": list(v.shape),
                              "sha256": hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()}
    json.dump(meta, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("wrote %s (%d entries, %.1f KiB)" % (path, len(g), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
