#!/usr/bin/env python3
"""tools/pcie_rate.py -- the chain's rate for a caller that holds HOST buffers ("PCIe").

Not bench.py's `value` (that one starts with the block batch resident in HBM).  Here every block batch is copied from
pinned host memory, processed, and the audio copied back, the three legs overlapped on three HIP streams with two device
buffers per direction -- what an application feeding the library from a host-side receiver front end would do.
Workload: the c4 shape (8192 SSB channels x 2^14 samples per block batch, fp32 chain, 2 B in + 4 B out per sample).

  python tools/pcie_rate.py [--batches 24] [--arith f32|q15]
prints one JSON line: copy-only rates of each direction, the kernel-only rate, and the overlapped end-to-end rate.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "minimal-sdr_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--arith", default="f32", choices=["f32", "q15"])
    args = ap.parse_args()
    import torch
    import msdr
    import bench

    if not torch.cuda.is_available():
        raise SystemExit("pcie_rate.py needs a GPU")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    wl = bench.workload("c4", msdr, 0)
    ch, n = wl["channels"], wl["n"]
    s_in, s_run, s_out = (torch.cuda.Stream(device=dev) for _ in range(3))
    ctx = msdr.Context(0, s_run.cuda_stream)
    q15 = args.arith == "q15"
    if q15:       # demodulation() only: the Teensy biquad nodes are serial per channel and would hide the link
        qi = [np.round(np.asarray(c, np.float64) * 32767).astype(np.int16) for c in wl["ci"]]
        qq = [np.round(np.asarray(c, np.float64) * 32767).astype(np.int16) for c in wl["cq"]]
        qo = [np.round(np.asarray(o, np.float64) * 32768).clip(-32768, 32767).astype(np.int16) for o in wl["osc"]]
        chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, qi, qq, mixer=wl["mixer"], mode=wl["mode"], osc_i=qo[0], osc_q=qo[1])
    else:
        chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, wl["ci"], wl["cq"], mixer=wl["mixer"], mode=wl["mode"],
                           osc_i=wl["osc"][0], osc_q=wl["osc"][1], biquad_coeffs=wl["bq"])
    odt = torch.int16 if q15 else torch.float32
    with torch.cuda.stream(s_run):
        seed = bench.synth_if(torch, dev, ch, n, wl["seed"])
    torch.cuda.synchronize(dev)
    h_in = [torch.empty((ch, n), dtype=torch.int16).pin_memory() for _ in range(2)]
    h_out = [torch.empty((ch, n), dtype=odt).pin_memory() for _ in range(2)]
    for h in h_in:
        h.copy_(seed.cpu())
    d_in = [torch.empty((ch, n), dtype=torch.int16, device=dev) for _ in range(2)]
    d_out = [torch.empty((ch, n), dtype=odt, device=dev) for _ in range(2)]
    K = args.batches
    samples = ch * n

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps

    def h2d():
        with torch.cuda.stream(s_in):
            d_in[0].copy_(h_in[0], non_blocking=True)

    def d2h():
        with torch.cuda.stream(s_out):
            h_out[0].copy_(d_out[0], non_blocking=True)

    def run():
        chain.process(d_in[0].data_ptr(), d_out[0].data_ptr(), n)

    t_h2d, t_d2h, t_run = timed(h2d, 8), timed(d2h, 8), timed(run, 8)

    # overlapped pipeline: batch k uses buffer k&1 on every leg
    ev_in = [torch.cuda.Event() for _ in range(2)]       # H2D of this buffer finished
    ev_run = [torch.cuda.Event() for _ in range(2)]      # kernel finished: d_in[b] free again, d_out[b] ready
    ev_out = [torch.cuda.Event() for _ in range(2)]      # D2H finished: d_out[b] free again

    def pipeline(batches):
        for k in range(batches):
            b = k & 1
            with torch.cuda.stream(s_in):
                if k >= 2:
                    s_in.wait_event(ev_run[b])
                d_in[b].copy_(h_in[b], non_blocking=True)
                ev_in[b].record(s_in)
            s_run.wait_event(ev_in[b])
            if k >= 2:
                s_run.wait_event(ev_out[b])
            chain.process(d_in[b].data_ptr(), d_out[b].data_ptr(), n)
            with torch.cuda.stream(s_run):
                ev_run[b].record(s_run)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_run[b])
                h_out[b].copy_(d_out[b], non_blocking=True)
                ev_out[b].record(s_out)

    pipeline(4)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    pipeline(K)
    torch.cuda.synchronize(dev)
    t_pipe = (time.perf_counter() - t0) / K
    in_b, out_b = 2 * samples, (2 if q15 else 4) * samples
    print(json.dumps({
        "what": "host-buffer caller, c4 shape, %s chain%s" % (args.arith, " (demodulation() only)" if q15 else ""),
        "channels": ch, "samples_per_channel": n, "batches": K,
        "h2d_GBps": round(in_b / t_h2d / 1e9, 2), "d2h_GBps": round(out_b / t_d2h / 1e9, 2),
        "kernel_only_Gsamples_s": round(samples / t_run / 1e9, 2),
        "end_to_end_Gsamples_s": round(samples / t_pipe / 1e9, 3),
        "end_to_end_ms_per_batch": round(t_pipe * 1e3, 3),
        "link_bound_Gsamples_s": round(1.0 / max(t_h2d, t_d2h) * samples / 1e9, 3),
        "kernel": chain.info()["kernel"],
    }))
    chain.close()
    ctx.close()


if __name__ == "__main__":
    main()
