"""Can the Teensy biquad (filter_biquad.cpp:33-82) be split in time by speculation -- start a segment from a guessed state after a
warm-up and keep it if its trajectory has merged with the true one?  It would have to merge BIT-EXACTLY (y history and the 14-bit
residue).  This study runs two trajectories of the oracle's model over the same input, one from the stream's start and one started
4096 samples later from zero state, and reports whether / when the full state (definition words 5..7) coincides and how often the
outputs differ.  Result (round 2): they never merge -- the residue difference performs a random walk over its 14 bits, and the
outputs differ by 1 LSB (LP) or up to 5 LSB (Q = 15 notch) in 35-70 % of the samples for as long as one cares to run.  Run:
    python tests/debug/biquad_merge_study.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import orclib  # noqa: E402

orc = orclib.Oracle()
CORR = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
FILTERS = {"LP 0.9*6 kHz Q 0.54 (biquad1_dac, .ino:391-393)": orc.biquad_design(orclib.BQ_LOWPASS, np.float32(6000 * 0.9 * CORR), 0.54),
           "notch fs/8 Q 15 (biquad2_dac, .ino:356)": orc.biquad_design(orclib.BQ_NOTCH, np.float32(24000 / 8 * CORR), 15.0)}


def study(coef, sig, warm_from=4096):
    a, b = orc.biquad_teensy_new([coef]), orc.biquad_teensy_new([coef])
    ya = orc.biquad_teensy_update(a, sig)
    yb = orc.biquad_teensy_update(b, sig[warm_from:])
    d = ya[warm_from:].astype(np.int64) - yb
    settled = d[2000:]
    # state equality, sample pair by sample pair
    a, b = orc.biquad_teensy_new([coef]), orc.biquad_teensy_new([coef])
    orc.biquad_teensy_update(a, sig[:warm_from])
    merged = None
    for k in range(warm_from, sig.size - 1, 2):
        orc.biquad_teensy_update(a, sig[k:k + 2])
        orc.biquad_teensy_update(b, sig[k:k + 2])
        if list(a.definition[5:8]) == list(b.definition[5:8]):
            merged = k - warm_from
            break
    return merged, float((settled != 0).mean()), int(np.abs(settled).max())


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    n = 1 << 16
    signals = {"uniform noise +-8000": rng.integers(-8000, 8001, n).astype(np.int16),
               "700 Hz tone + noise": (6000 * np.cos(2 * np.pi * 700 * np.arange(n) / 24000) + rng.integers(-300, 301, n)).astype(np.int16),
               "noise +-50": rng.integers(-50, 51, n).astype(np.int16)}
    for fname, coef in FILTERS.items():
        for sname, sig in signals.items():
            merged, frac, worst = study(coef, sig)
            print("%-48s %-22s merged after: %-6s outputs differ in %.1f %% of samples (max %d LSB)" % (fname, sname, merged, 100 * frac, worst))
