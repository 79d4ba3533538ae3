import sys, os, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, 'minimal-sdr_amd/python')
import orclib, msdr
from test_gpu_chain import _hilbert_pair, _nco, _f32_biquads, run_chain, rel_rms
ctx = msdr.Context(0); orc = orclib.Oracle()
for ntaps in (240, 256, 300, 384, 512):
    for mode in (orclib.AM, orclib.LSB):
        rng = np.random.default_rng(ntaps * 3 + mode)
        if mode == orclib.AM:
            hi = (np.sinc(2 * 2800 / 24000 * (np.arange(ntaps) - (ntaps - 1) / 2)) * np.kaiser(ntaps, 7.0)).astype(np.float32); hi /= hi.sum(); hq = hi
        else:
            hi, hq = _hilbert_pair(ntaps)
        oi, oq = _nco(128, 5)
        bq = _f32_biquads(orc, 2)
        x = rng.integers(-8000, 8001, (3, 20011)).astype(np.int16)
        chain = msdr.Chain(ctx, msdr.ARITH_F32, 3, hi, hq, mixer=msdr.MIXER_NCO, mode=mode, osc_i=oi, osc_q=oq, biquad_coeffs=bq)
        for block in (None, 4999):
            chain.reset()
            got = run_chain(ctx, chain, x, np.float32, block)
            errs = [rel_rms(got[c], orc.chain_f32(x[c], mode, hi, hq, oi, oq, bq)) for c in range(3)]
            print(ntaps, mode, block, chain.info()["kernel"], chain.info()["lds_bytes"], chain.info()["block"], "max err %.2e" % max(errs), flush=True)
