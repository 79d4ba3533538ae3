#!/usr/bin/env python3
"""bench.py -- throughput of the demodulation path on MI355X (one JSON line on rank 0).

A "step" is ONE pass of the hot path over one block batch of synthetic IF that is already resident in HBM.

Default (no --workload): the HEADLINE is c3 = BASELINE.json configs[2], the north-star's "HBM-roofline run" (4096 AM channels,
256-tap fp32 FIR pair, Fs/4 mix, envelope, 2-stage biquad; int16 in, fp32 out), timed as the contract says (W warm-up steps,
exactly K steps between barriers, MAX over ranks).  The same line carries, under "also", the other single-GPU records of the
path, each with its own timing, roofline and parity:
  fir   the 256-tap fp32 FIR STAGE alone (arm_fir_f32, 8 B per sample) -- the stage the north-star's 70 % target names
  c2    1 SSB channel, NCO tables at fs/4, 100-tap Hilbert pair, LSB, 2^30 int16 samples          [configs[1]]
  c4    8192 SSB channels per GPU x 2^14 samples, 100-tap pair, LSB                               [configs[3]]
  c5    256 mixed AM/LSB channels per GPU x 2^20 samples, 512 taps                                [configs[4]]
--workload X runs one record alone (c2..c5, fir, plus fe = front end, spec = spectrum FFT).
Parity is checked inside the run against the CPU oracle (test infrastructure, never the thing measured): the head of the
stream from zero state AND windows that straddle the kernel's time-segment boundaries and the tail, each with an oracle
pre-roll (as tests/test_gpu_fullsize.py does).
Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL).  Channels are independent, so every rank runs its own
shard with NO data-path collective ("weak" scaling); the RCCL gather of demodulated audio is timed separately, outside the
timed region (all-gather, gather-to-root, and the double-buffered schedule that overlaps gather k with compute k + 1), and
reported under "gather".  The CPU baseline runs on rank 0 whatever N is.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "minimal-sdr_amd", "python"))

FS = 24000.0
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GUIDE_COPY_GBS = 6290.0        # MI355X_MICROARCH.md:36: "6.29 TB/s measured (float4 copy, 79%)"
MEASURED_COPY_GBS = 6590.0     # this repo's own best 1:1 read / write stream on the part: one-shot float4 copy, 4 GiB -> 4 GiB, non-temporal loads and
                               # stores, 1.303 ms (profiles/r03/stream_order.md, tools/probes/copy_ceiling_probe.hip)


def ceiling_fracs(roofline):
    """Adds the fraction against the guide's measured copy rate (SURVEY 8d asks for both denominators) and against the best copy this
    repo measured itself; `frac` stays achieved / 8 TB/s."""
    a = roofline["achieved"]
    roofline["frac_vs_6.29TBps"] = round(a / GUIDE_COPY_GBS, 4)
    roofline["frac_of_measured_copy"] = round(a / MEASURED_COPY_GBS, 4)
    roofline["measured_copy_GBps"] = MEASURED_COPY_GBS
    return roofline
MFMA_F16_SUSTAINED_TFLOPS = 1250.0   # MI355X_MICROARCH.md, DVFS item 1: what a tuned fp16 GEMM sustains on random data at the 1400 W cap
RIDGE_FLOP_PER_BYTE = 2500.0e12 / 8000.0e9      # 312.5: past it a record is matrix-core-bound, not HBM-bound


def say_which_roof(roofline, alg_bytes_per_sample):
    """Names the roof that binds a matrix-core record: executed fp16 flop per ALGORITHMIC byte above the ridge (2.5 PF / 8 TB/s = 312) means
    the matrix cores bind, and the yardstick is the sustained fp16 rate (1.25 PF under DVFS), not the HBM fraction.  `frac` stays
    achieved / 8 TB/s for every record (the contract's figure); `bound` and `binding_frac` say which number to read."""
    tf = roofline.get("mfma_f16_tflops_executed")
    if tf is None or not roofline.get("achieved"):
        return roofline
    fpb = tf * 1e12 / (roofline["achieved"] * 1e9)             # flop per algorithmic byte (both per second of the same kernel)
    roofline["mfma_flop_per_alg_byte"] = round(fpb, 1)
    roofline["ridge_flop_per_byte"] = round(RIDGE_FLOP_PER_BYTE, 1)
    roofline["mfma_frac_of_sustained"] = round(tf / MFMA_F16_SUSTAINED_TFLOPS, 4)
    roofline["mfma_sustained_tflops"] = MFMA_F16_SUSTAINED_TFLOPS
    if fpb > 1.05 * RIDGE_FLOP_PER_BYTE:                        # (c3 sits AT the ridge, 320 against 312: its record keeps the HBM yardstick the metric names)
        roofline["bound"] = "mfma_f16"
        roofline["binding_frac"] = roofline["mfma_frac_of_sustained"]
    else:
        roofline["binding_frac"] = roofline["frac"]
    return roofline


def compact_record(rec):
    """What the one JSON line keeps of a sub-record (the driver parses top-level keys and keeps `roofline` whole; the full record,
    parity windows included, goes to bench_full.json)."""
    if rec is None:
        return None
    r, par = rec.get("roofline", {}), rec.get("parity") or {}
    # what the parity figure is, by its key: relative RMS error against the oracle (fp32), mismatching samples (integer paths: 0 = bit-exact),
    # or the largest difference in int16 steps (int16 audio out of the fp32 chain: tolerance 1)
    pkey = "rel_rms_worst" if "rel_rms_worst" in par else "mismatching_samples" if "mismatching_samples" in par else "max_abs_lsb"
    c = {"Msps": rec.get("value"), "ms_per_step": rec.get("ms_per_step"), "kernel_ms": r.get("kernel_ms"), "frac": r.get("frac"),
         "bound": r.get("bound"), "sclk_mhz": r.get("sclk_mhz"), "power_w": r.get("power_w"),
         {"rel_rms_worst": "parity", "mismatching_samples": "parity_mismatches", "max_abs_lsb": "parity_max_lsb"}[pkey]: par.get(pkey),
         "kernel": str(rec.get("config", {}).get("kernel", ""))[:40]}
    for k in ("binding_frac", "mfma_frac_of_sustained", "step_ms", "node_pass_ms", "traffic", "tick_us", "launches_per_step", "graph_tick_us"):
        if r.get(k) is not None:
            c[k] = r[k]
    return c


def power_probe(step, torch, dev, device_index, world):
    """Shader clock and socket power while the record's step loops UNTIMED (after the timed region): every matrix-core kernel of this
    bench runs at the package's power cap, where time = energy / cap and the clock is whatever the cap leaves -- the figure that explains
    a roofline fraction belongs next to it.  Three rocm-smi samples, ~0.3 s apart, taken while a helper thread keeps the queue fed.
    Single-GPU runs only (rank 0 would otherwise hold the other ranks at the next barrier)."""
    if world != 1:
        return {"sclk_mhz": None, "power_w": None, "power_note": "sampled on single-GPU runs only"}
    if os.environ.get("MSDR_BENCH_NO_POWER", "0") == "1":       # profiler runs (tools/profile.sh): no helper thread, no child process
        return {"sclk_mhz": None, "power_w": None, "power_note": "not sampled (MSDR_BENCH_NO_POWER=1)"}
    import re
    import subprocess
    import threading
    stop = threading.Event()

    def feed():
        while not stop.is_set():
            for _ in range(8):
                step()
            torch.cuda.synchronize(dev)
    th = threading.Thread(target=feed, daemon=True)
    th.start()
    time.sleep(0.6)
    clocks, watts, err = [], [], None
    for _ in range(3):
        try:
            txt = subprocess.run(["rocm-smi", "-d", str(device_index), "--showclocks", "--showpower"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 text=True, timeout=20).stdout
            m = re.search(r"sclk clock level:[^(]*\((\d+)Mhz\)", txt)
            w = re.search(r"Socket[^:]*Power \(W\):\s*([0-9.]+)", txt)
            if m:
                clocks.append(int(m.group(1)))
            if w:
                watts.append(float(w.group(1)))
        except (OSError, subprocess.SubprocessError) as e:
            err = str(e)[:120]
        time.sleep(0.3)
    stop.set()
    th.join()
    torch.cuda.synchronize(dev)
    out = {"sclk_mhz": int(np.median(clocks)) if clocks else None, "power_w": float(np.median(watts)) if watts else None,
           "power_samples": {"sclk_mhz": clocks, "power_w": watts, "how": "rocm-smi --showclocks --showpower, 3 samples while the step loops untimed after the timed region"}}
    if err and not clocks:
        out["power_note"] = "rocm-smi unavailable: " + err
    return out


VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 matrix peak (same guide)


def hilbert_pair(n_taps, fc=1330.0, bw=1920.0):
    k = np.arange(n_taps)
    m = (n_taps - 1) / 2.0
    proto = np.sinc(bw / FS * (k - m)) * np.kaiser(n_taps, 6.0)
    proto /= proto.sum()
    w = 2 * np.pi * fc / FS
    return ((2 * proto * np.cos(w * (k - m) + np.pi / 4)).astype(np.float32),
            (2 * proto * np.cos(w * (k - m) - np.pi / 4)).astype(np.float32))


def lowpass(n_taps, fc=2800.0):
    k = np.arange(n_taps)
    h = np.sinc(2 * fc / FS * (k - (n_taps - 1) / 2.0)) * np.kaiser(n_taps, 7.0)
    return (h / h.sum()).astype(np.float32)


def lowpass_designer(msdr, n_taps, fc=2800.0):
    """The low-pass as the REFERENCE designs it: calc_FIR_coeffs(FIR_AM_coeffs, n, filter_bandwidth, 70 dB, 0, 0, 24000) (calc_demod_filter,
    Minimal-SDR.ino:221-223; designer :782-872, restated bit-exactly by the library's msdr_calc_FIR_coeffs), q15 taps converted as
    arm_q15_to_float does.  Linear phase about an integer index (c[k] = c[n - k]): what chain_amsy_kernel's folded window needs."""
    return (msdr.calc_fir_coeffs(n_taps, fc, 70.0, 0, 0.0, FS)[:n_taps].astype(np.float32) / 32768.0).astype(np.float32)


def reference_biquads(msdr):
    """biquad1_dac: LP 0.9*6 kHz Q=0.54 (.ino:391-393); biquad2_dac: notch fs/8 Q=15 (.ino:356); CMSIS sign."""
    corr = msdr.AUDIO_SAMPLE_RATE_EXACT / FS
    out = []
    for kind, f, q in ((msdr.BQ_LOWPASS, 6000 * 0.9 * corr, 0.54), (msdr.BQ_NOTCH, FS / 8 * corr, 15.0)):
        c = msdr.biquad_design(kind, np.float32(f), q).astype(np.float64) / 2 ** 30
        out.append([c[0], c[1], c[2], -c[3], -c[4]])
    return np.array(out, np.float32)


def workload(name, msdr, rank, osc_period=4, lp_design="numpy"):
    # freq_conv-style oscillator: the reference's tables are q15 (Osc_I/Q_buffer_i, freq_conv.h:33-34), one
    # AUDIO_BLOCK long, here at fs/4; converted to float as arm_q15_to_float does (/32768)
    osc_n = np.arange(128)
    osc_i = (np.round(32767 * np.sin(2 * np.pi * osc_n / osc_period)).astype(np.int16) / 32768.0).astype(np.float32)
    osc_q = (np.round(32767 * np.cos(2 * np.pi * osc_n / osc_period)).astype(np.int16) / 32768.0).astype(np.float32)
    bq = reference_biquads(msdr)
    if name == "c2":
        hi, hq = hilbert_pair(100)
        return dict(name="c2: 1 SSB channel, NCO mix + 100-tap Hilbert pair + LSB + 2-stage biquad, 2^30 int16 IF",
                    channels=1, n=1 << 30, taps=100, ci=[hi], cq=[hq], mixer=msdr.MIXER_NCO, osc=(osc_i, osc_q),
                    modes=None, tapsets=None, mode=msdr.MODE_LSB, bq=bq, seed=2 + 1000 * rank)
    if name == "c3":
        lp = lowpass_designer(msdr, 256) if lp_design == "designer" else lowpass(256)
        return dict(name="c3: 4096 AM channels x 2^18, Fs/4 mix + 256-tap low-pass pair + envelope + 2-stage biquad",
                    channels=4096, n=1 << 18, taps=256, ci=[lp], cq=[lp], mixer=msdr.MIXER_FS4, osc=None,
                    modes=None, tapsets=None, mode=msdr.MODE_AM, bq=bq, seed=3 + 1000 * rank)
    if name == "c4":
        hi, hq = hilbert_pair(100)
        return dict(name="c4: 8192 SSB channels/GPU x 2^14, NCO mix + 100-tap Hilbert pair + LSB + 2-stage biquad",
                    channels=8192, n=1 << 14, taps=100, ci=[hi], cq=[hq], mixer=msdr.MIXER_NCO, osc=(osc_i, osc_q),
                    modes=None, tapsets=None, mode=msdr.MODE_LSB, bq=bq, seed=4 + 1000 * rank)
    if name == "c5":
        ch = 256
        lp = lowpass_designer(msdr, 512) if lp_design == "designer" else lowpass(512)
        hi, hq = hilbert_pair(512)
        modes = np.array([msdr.MODE_AM if ((c + ch * rank) * 2654435761 >> 7) & 1 else msdr.MODE_LSB for c in range(ch)], np.int32)
        tapsets = np.array([0 if m == msdr.MODE_AM else 1 for m in modes], np.int32)
        return dict(name="c5: 256 mixed AM/LSB channels/GPU x 2^20, Fs/4 mix + 512-tap pair + 2-stage biquad",
                    channels=ch, n=1 << 20, taps=512, ci=[lp, hi], cq=[lp, hq], mixer=msdr.MIXER_FS4, osc=None,
                    modes=modes, tapsets=tapsets, mode=msdr.MODE_AM, bq=bq, seed=5 + 1000 * rank)
    raise SystemExit("unknown workload " + name)


def synth_if(torch, dev, channels, n, seed):
    """Two tones (6.7 kHz, 5.4 kHz at -6 dB) + uniform noise +-500, int16, generated on the device."""
    x = torch.empty((channels, n), dtype=torch.int16, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    flat = x.view(-1)
    total = channels * n
    step = 1 << 24
    for o in range(0, total, step):
        m = min(step, total - o)
        t = ((torch.arange(o, o + m, device=dev, dtype=torch.int64) % n) % 240000).to(torch.float32)   # both tones have a 240000-sample period
        v = 6000.0 * torch.cos(t * (2 * math.pi * 6700.0 / FS)) + 3000.0 * torch.cos(t * (2 * math.pi * 5400.0 / FS) + 0.3)
        v += torch.randint(-500, 501, (m,), device=dev, generator=g).to(torch.float32)
        flat[o:o + m] = v.round().to(torch.int16)
    return x


def pick_team(run, xs, modes, ts, limit):
    """Size of the OpenMP team for the CPU leg: the team with the best rate over a ladder of team sizes
    up to `limit` (= usable_cores()), measured on a short weak-scaling sample (two rows of 2^16 samples per thread).  A lease that
    is a share of the machine -- by a quota this process cannot see -- shows up here as a rate that stops growing; the record's
    `cores` is the team that actually ran, never the machine's core count.  Returns (team, {team: Msamples/s})."""
    ladder = sorted({c for c in (limit, 192, 128, 96, 64, 48, 32, 24, 16, 12, 8, 4, 2, 1) if c <= limit and c <= xs.shape[0]}, reverse=True) or [1]
    cols = min(xs.shape[1], 1 << 16)
    rates = {}
    for t in ladder:
        rows = min(xs.shape[0], 2 * t)
        sub = np.ascontiguousarray(xs[:rows, :cols])
        best = 0.0
        for _ in range(2):                                     # the first pass of a new team also starts its threads
            _, dt, _ = run(sub, modes[:rows], ts[:rows], t)
            best = max(best, rows * cols / dt)
        rates[t] = best
    top = max(rates.values())
    team = min(t for t, r in rates.items() if r >= top)        # the team with the BEST rate (ties: the smaller one)
    return team, {str(t): round(r / 1e6, 2) for t, r in sorted(rates.items())}


def cpu_baseline(wl, x_host, gpu_first, target_s=12.0):
    """Time the oracle (CPU restatement, "port") on a bounded sample of the SAME input: the first
    samples of up to `threads` channels; one-channel workloads are cut into per-thread time chunks
    (fresh zero state per chunk: identical arithmetic work per sample).  Also returns parity of the
    GPU output against the oracle on that sample (first chunk of each channel only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    orc = orclib.Oracle()
    threads = usable_cores()[0]
    osc_i, osc_q = wl["osc"] if wl["osc"] else (np.array([0, 1, 0, -1], np.float32), np.array([1, 0, -1, 0], np.float32))
    modes_all = wl["modes"] if wl["modes"] is not None else np.full(wl["channels"], wl["mode"], np.int32)
    tapsets_all = wl["tapsets"] if wl["tapsets"] is not None else np.zeros(wl["channels"], np.int32)

    def run(xs, modes, ts, thr):
        # one tap set per oracle call: group rows by tap set; only the oracle's own batch driver is timed (not numpy's row copies)
        outs = np.empty(xs.shape, np.float32)
        used, dt = 1, 0.0
        for s in sorted(set(ts.tolist())):
            idx = np.nonzero(ts == s)[0]
            xi, mi = (xs, modes) if idx.size == xs.shape[0] else (np.ascontiguousarray(xs[idx]), modes[idx])
            t0 = time.perf_counter()
            o, used = orc.chain_f32_batch(xi, mi, wl["ci"][s], wl["cq"][s], osc_i, osc_q, wl["bq"] if len(wl["bq"]) else None, threads=thr)
            dt += time.perf_counter() - t0
            outs[idx] = o
        return outs, dt, used

    # 1-thread rate on a short prefix
    cal_n = min(x_host.shape[1], 200000)
    _, dt1, _ = run(x_host[:1, :cal_n], modes_all[:1], tapsets_all[:1], 1)
    rate1 = cal_n / dt1
    if x_host.shape[0] == 1:                                  # one channel: cut the sample into per-thread time chunks
        chunks = max(1, min(threads, x_host.shape[1] // 65536))
        per_row = x_host.shape[1] // chunks
        xs = np.ascontiguousarray(x_host[0, :chunks * per_row].reshape(chunks, per_row))
        modes, ts = np.full(chunks, modes_all[0], np.int32), np.full(chunks, tapsets_all[0], np.int32)
    else:
        per_row = x_host.shape[1]
        xs, modes, ts = x_host, modes_all[:x_host.shape[0]], tapsets_all[:x_host.shape[0]]
    threads, ladder = pick_team(run, xs, modes, ts, threads)
    total_dt, passes, outs, used = 0.0, 0, None, 1
    while total_dt < target_s and passes < 5000:                # repeat whole passes until ~target_s of wall time
        outs, dt, used = run(xs, modes, ts, threads)
        total_dt += dt
        passes += 1
    rate = xs.size * passes / total_dt
    # parity on rows that start at stream position 0 with zero state
    par_rows = [0] if x_host.shape[0] == 1 else list(range(xs.shape[0]))
    worst = 0.0
    for r in par_rows:
        m = min(per_row, gpu_first.shape[1])
        want, got = outs[r, :m].astype(np.float64), gpu_first[r, :m].astype(np.float64)
        worst = max(worst, float(np.sqrt(((got - want) ** 2).sum() / max((want ** 2).sum(), 1e-300))))
    cores = int(min(used, xs.shape[0]))
    return {"value": round(rate / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "one_thread_Msamples_per_s": round(rate1 / 1e6, 3), "scaling_vs_one_thread": round(rate / (cores * rate1), 3),
            "team_ladder_Msamples_per_s": ladder,
            "sample": "%d row(s) x %d samples of the same IF input, %d pass(es), %.1f s wall (oracle/msdr_oracle.c orc_chain_f32; OpenMP team of %d = "
                      "the team with the best rate on this lease, see team_ladder)" % (xs.shape[0], per_row, passes, total_dt, cores)}, worst, min(per_row, gpu_first.shape[1])


def cpu_baseline_q15(wl, x_host, gpu_first, target_s=12.0):
    """Same as cpu_baseline for the integer chain (orc_chain_q15_batch); parity = number of differing samples."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    orc = orclib.Oracle()
    threads = usable_cores()[0]
    rows = x_host.shape[0]
    n = (x_host.shape[1] // 128) * 128
    modes_all = wl["modes"] if wl["modes"] is not None else np.full(wl["channels"], wl["mode"], np.int32)
    tapsets_all = wl["tapsets"] if wl["tapsets"] is not None else np.zeros(wl["channels"], np.int32)
    nodes = [orc.biquad_teensy_new(c) for c in wl["qnodes"]]
    oi, oq = wl["qosc"] if wl["qosc"] else (None, None)

    def run(xs, modes, ts, thr):
        outs = np.empty(xs.shape, np.int16)
        used, dt = 1, 0.0
        for s_ in sorted(set(ts.tolist())):
            idx = np.nonzero(ts == s_)[0]
            xi, mi = (xs, modes) if idx.size == xs.shape[0] else (np.ascontiguousarray(xs[idx]), modes[idx])
            t0 = time.perf_counter()
            o, used = orc.chain_q15_batch(xi, mi, wl["qi"][s_], wl["qq"][s_], mixer=1 if oi is not None else 0,
                                          osc_i=oi, osc_q=oq, biquads=nodes, threads=thr)
            dt += time.perf_counter() - t0
            outs[idx] = o
        return outs, dt, used

    if rows == 1:
        chunks = max(1, min(threads, n // 65536))
        per_row = (n // chunks // 128) * 128
        xs = np.ascontiguousarray(x_host[0, :chunks * per_row].reshape(chunks, per_row))
        modes, ts = np.full(chunks, modes_all[0], np.int32), np.full(chunks, tapsets_all[0], np.int32)
    else:
        per_row = n
        xs, modes, ts = np.ascontiguousarray(x_host[:, :n]), modes_all[:rows], tapsets_all[:rows]
    threads, ladder = pick_team(run, xs, modes, ts, threads)
    cal = np.ascontiguousarray(xs[:1, :min(per_row, 1 << 17)])
    _, dt1, _ = run(cal, modes[:1], ts[:1], 1)
    rate1 = cal.size / dt1
    total_dt, passes, outs, used = 0.0, 0, None, 1
    while total_dt < target_s and passes < 5000:
        outs, dt, used = run(xs, modes, ts, threads)
        total_dt += dt
        passes += 1
    rate = xs.size * passes / total_dt
    m = min(per_row, gpu_first.shape[1])
    bad = 0
    for r in ([0] if rows == 1 else range(xs.shape[0])):
        bad += int((outs[r, :m] != gpu_first[r, :m]).sum())
    cores = int(min(used, xs.shape[0]))
    return {"value": round(rate / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "one_thread_Msamples_per_s": round(rate1 / 1e6, 3), "scaling_vs_one_thread": round(rate / (cores * rate1), 3),
            "team_ladder_Msamples_per_s": ladder,
            "sample": "%d row(s) x %d samples of the same IF input, %d pass(es), %.1f s wall (oracle/msdr_oracle.c orc_chain_q15; OpenMP team of %d)"
                      % (xs.shape[0], per_row, passes, total_dt, cores)}, bad, m


def bench_frontend(args, torch, msdr, ctx, dev, rank, world, dist):
    """Row f1: DC block + AudioAmplifier + AGC over 4096 channels x 2^18 raw conversions (integer, bit-exact).  One lane per
    channel (the recurrences are exact only in order), so this is latency-bound by construction; reported for completeness."""
    ch, n = args.channels or 4096, args.samples or (1 << 18)
    g = torch.Generator(device=dev)
    g.manual_seed(11 + rank)
    t = torch.arange(n, device=dev, dtype=torch.float32)
    x = (32768 + 6000 * torch.cos(t * (2 * math.pi * 6000.0 / FS))[None, :] * (0.3 + torch.rand((ch, 1), device=dev, generator=g))
         + torch.randint(-60, 61, (ch, n), device=dev, generator=g)).clamp(0, 65535).to(torch.int32)
    x = (x - 65536 * (x >= 32768).to(torch.int32)).to(torch.int16)          # the same 16 bits, viewed as int16 storage
    y = torch.empty((ch, n), dtype=torch.int16, device=dev)
    fe = msdr.Frontend(ctx, ch)
    fe.prime(np.uint16(32768))
    torch.cuda.synchronize(dev)
    fe.update(x.data_ptr(), y.data_ptr(), n)
    torch.cuda.synchronize(dev)
    first = y[:8].cpu().numpy() if rank == 0 else None
    for _ in range(max(0, args.warmup - 1)):
        fe.update(x.data_ptr(), y.data_ptr(), n)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fe.update(x.data_ptr(), y.data_ptr(), n)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import msdr_dist
        dt = msdr_dist.max_over_ranks(dt, args.cdev)
    if rank != 0:
        return None
    value = world * ch * n * args.steps / dt / 1e6
    ms = dt / args.steps * 1e3
    out = {"metric": "Msamples/s through the front end (DC block -> AudioAmplifier -> AGC); achieved HBM GB/s vs peak",
           "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "q15 (uint16 in, int16 out, int32/int64 recurrences)", "data": "synthetic",
           "config": {"workload": "fe: %d channels x %d raw conversions, DC block + gain + AGC per 128-sample block" % (ch, n),
                      "channels_per_gpu": ch, "samples_per_channel_per_step": n, "kernel": "frontend_pipe4_kernel"},
           "roofline": {"bound": "hbm", "achieved": round(4.0 * ch * n / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(4.0 * ch * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                        "note": "one lane per channel: serial integer recurrences, latency-bound by construction"}}
    if not (args.channels or args.samples):
        # The same number of conversions as 4x the channels: a workgroup (64 channels) walks its stream alone, so 4096 channels keep 64 of the
        # 256 CUs busy and the step time is the per-channel latency (2^18 dependent recursion steps); 16 384 channels fill the chip.
        chw, nw = 4 * ch, n // 4
        xw, yw = x.view(-1)[:chw * nw].view(chw, nw), y.view(-1)[:chw * nw].view(chw, nw)     # (the values do not matter for the time)
        few = msdr.Frontend(ctx, chw)
        few.prime(np.uint16(32768))
        for _ in range(3):
            few.update(xw.data_ptr(), yw.data_ptr(), nw)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            few.update(xw.data_ptr(), yw.data_ptr(), nw)
        torch.cuda.synchronize(dev)
        msw = (time.perf_counter() - t1) / args.steps * 1e3
        few.close()
        out["wide"] = {"workload": "fe: %d channels x %d raw conversions (the same amount of data, 4x the channels)" % (chw, nw),
                       "ms_per_step": round(msw, 4), "value": round(chw * nw / (msw * 1e-3) / 1e6, 1),
                       "roofline_frac": round(4.0 * chw * nw / (msw * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orclib
        orc = orclib.Oracle()
        xs = x[:8, :1 << 16].cpu().numpy().view(np.uint16)
        t1 = time.perf_counter()
        bad = 0
        for c in range(xs.shape[0]):
            f = orc.frontend_new(first_conversion=32768)
            want = orc.frontend_run(f, xs[c])
            bad += int((want != first[c, :xs.shape[1]]).sum())
        cdt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(xs.size / cdt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
                               "sample": "%d channels x %d samples of the same input (oracle/msdr_oracle.c orc_frontend_run)" % xs.shape}
        out["parity"] = {"mismatching_samples": bad, "tolerance": 0}
    ceiling_fracs(out["roofline"])
    return out


def bench_spectrum(args, torch, msdr, ctx, dev, rank, world, dist):
    """Row f4: the display's 128-point q15 real FFT + column heights (UI.cpp:520-592) over every 128-sample block of a
    4096-channel batch (the reference transforms one block in 25; here all of them, as a throughput figure)."""
    ch, n = args.channels or 4096, args.samples or (1 << 15)
    nfft = ch * (n // 128)
    g = torch.Generator(device=dev)
    g.manual_seed(13 + rank)
    x = torch.randint(-20000, 20001, (nfft, 128), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
    o = torch.empty((nfft, 256), dtype=torch.int16, device=dev)
    c = torch.empty((nfft, 128), dtype=torch.uint8, device=dev)
    lib = ctx.lib
    import ctypes as C

    def step():
        rc = lib.msdr_rfft128_q15(ctx.h, C.c_void_p(x.data_ptr()), C.c_uint64(128), C.c_void_p(o.data_ptr()), C.c_void_p(c.data_ptr()),
                                  C.c_uint32(nfft))
        if rc != 0:
            raise SystemExit("msdr_rfft128_q15: %s" % lib.msdr_last_error().decode())
    warm_run = 0
    t_w = time.perf_counter()
    while warm_run < max(1, args.warmup) or time.perf_counter() - t_w < 0.2:     # 0.2 ms steps: warm up by time as well (see timed_steps)
        step()
        warm_run += 1
        if warm_run % 64 == 0:
            torch.cuda.synchronize(dev)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import msdr_dist
        dt = msdr_dist.max_over_ranks(dt, args.cdev)
    if rank != 0:
        return None
    ms = dt / args.steps * 1e3
    gbs = 896.0 * nfft / (ms * 1e-3) / 1e9
    out = {"metric": "Msamples/s through the spectrum FFT (arm_rfft_q15 128 points + column heights); achieved HBM GB/s vs peak",
           "value": round(world * nfft * 128 * args.steps / dt / 1e6, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "warmup_steps_run": warm_run, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "q15 (int16 in/out, 32-bit wrap-around products)", "data": "synthetic",
           "config": {"workload": "spec: %d transforms of 128 samples (%d channels x %d blocks)" % (nfft, ch, n // 128),
                      "kernel": "spectrum_rfft128_kernel"},
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "traffic": None, "note": "896 B per transform: 256 in + 512 FFT_out + 128 column bytes"}}
    if not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orclib
        orc = orclib.Oracle()
        k = 4096
        xs, og, cg = x[:k].cpu().numpy(), o[:k].cpu().numpy(), c[:k].cpu().numpy()
        t1 = time.perf_counter()
        bad = 0
        for f in range(k):
            want, _ = orc.rfft128_q15(xs[f])
            bad += int((want != og[f]).sum()) + int((orc.spectrum_columns(want) != cg[f, :127]).sum())
        cdt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(k * 128 / cdt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
                               "sample": "%d transforms of the same input through ctypes (oracle/msdr_oracle.c orc_rfft128_q15)" % k}
        out["parity"] = {"mismatching_values": bad, "tolerance": 0}
    ceiling_fracs(out["roofline"])
    return out


def usable_cores():
    """How many cores this process may really use: the scheduler affinity mask, cut down to the cgroup's CPU quota where one is
    set (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  os.cpu_count() is the machine's count, not the lease's: on
    the GPU box it says 256 while a one-GPU lease is a share of those.  Returns (cores, affinity, quota or None)."""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    cores = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return cores, aff, quota


def host_info():
    """CPU model and core counts of the box the CPU baseline runs on: the machine's, and what this process may use."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    cores, aff, quota = usable_cores()
    return {"nproc": os.cpu_count() or 1, "affinity_cores": aff, "cgroup_cpu_quota": quota, "usable_cores": cores, "cpu_model": model}


def timed_steps(args, torch, dev, dist, step, after_warmup=None):
    """The contract's timed region: W untimed warm-up steps, then exactly K steps between barrier + synchronize pairs;
    returns the MAX over ranks of the wall time."""
    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    barrier()
    # sub-records ("also") only: the card drops to its idle clock (157 MHz here) while the host prepares a record, and W short steps
    # (c4: 0.24 ms each) end before it is back at its working clock -- keep stepping, untimed, until min_warm_s have passed
    extra, t_w = 0, time.perf_counter()
    while getattr(args, "min_warm_s", 0.0) > 0.0 and time.perf_counter() - t_w < args.min_warm_s:
        for _ in range(8):
            step()
        torch.cuda.synchronize(dev)
        extra += 8
    args.warmup_steps_run = args.warmup + extra
    barrier()
    if after_warmup is not None:
        after_warmup()                                           # e.g. clear the kernel-event timers: only the K timed steps count
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import msdr_dist
        dt = msdr_dist.max_over_ranks(dt, args.cdev)
    return dt


def bench_fir_stage(args, torch, msdr, ctx, dev, rank, world, dist, do_cpu):
    """The FIR STAGE on its own (rows A4 / A6): arm_fir_f32 (fp32 in / fp32 out, 8 B per sample) or, with --arith q15,
    arm_fir_fast_q15 (int16 in / out, 4 B per sample), 256 taps, batched over 4096 channels x 2^18 samples."""
    ch, n, nt = args.channels or 4096, args.samples or (1 << 18), args.taps or 256
    q15 = args.arith == "q15"
    g = torch.Generator(device=dev)
    g.manual_seed(17 + rank)
    lp = lowpass(nt)
    if q15:
        taps = np.round(lp.astype(np.float64) * 32767).astype(np.int16)
        x = torch.randint(-8000, 8001, (ch, n), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
        y = torch.empty((ch, n), dtype=torch.int16, device=dev)
        fir = msdr.FirQ15(ctx, taps, ch)
        fn = ctx.lib.msdr_fir_q15_process
    else:
        taps = lp
        x = (torch.rand((ch, n), device=dev, generator=g) * 16000.0 - 8000.0)
        y = torch.empty((ch, n), dtype=torch.float32, device=dev)
        fir = msdr.FirF32(ctx, taps, ch)
        fn = ctx.lib.msdr_fir_f32_process
    trn = "chain_q15mf_kernel<3> (i8 matrix cores)" if q15 else fir.kernel_name() + " (split-fp16 matrix cores)"
    import ctypes as C

    def step():
        if fn(fir.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_uint32(n)) != 0:
            raise SystemExit("fir process: %s" % ctx.lib.msdr_last_error().decode())
    step()                                                       # first pass from zero state: kept for the parity check
    torch.cuda.synchronize(dev)
    # parity samples: head, the middle of the block (inside another time segment of the kernel), and the tail, on four channels
    rows = sorted({0, 1, ch // 2, ch - 1})
    L = min(n, 1 << 14)
    los = sorted({0, max(0, n // 2 - L // 2), n - L})
    pre = 2 * nt
    keep = {(r, lo): (x[r, max(0, lo - pre):lo + L].cpu().numpy(), y[r, lo:lo + L].cpu().numpy()) for r in rows for lo in los} if rank == 0 else None
    ctx.enable_kernel_timing(True)
    dt = timed_steps(args, torch, dev, dist, step, after_warmup=ctx.kernel_time)
    k_total, launches = ctx.kernel_time()
    ctx.enable_kernel_timing(False)
    power = power_probe(step, torch, dev, dev.index or 0, world) if rank == 0 else None
    x_cpu = None
    if rank == 0 and do_cpu:                                     # rows of the same input for the timed CPU leg (two per thread of its team)
        x_cpu = x[:min(ch, max(2 * usable_cores()[0], 16)), :min(n, 1 << 18)].cpu().numpy()
    fir.close()
    del x, y
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    ms = dt / args.steps * 1e3
    k_ms = k_total / max(1, launches)
    bps = 4.0 if q15 else 8.0
    gbs = bps * ch * n / (k_ms * 1e-3) / 1e9
    out = {"metric": "Msamples/s through the %d-tap FIR stage alone (%s); achieved HBM GB/s vs peak" % (nt, "arm_fir_fast_q15" if q15 else "arm_fir_f32"),
           "value": round(world * ch * n * args.steps / dt / 1e6, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "q15 (int16 data, wrapping int32 accumulate)" if q15 else "f32 (fp32 accumulate; operands as 2 x fp16 pieces = 22 bits, block floating point per tile)",
           "data": "synthetic",
           "config": {"workload": "fir: %d channels x %d samples, %d taps, stage mirror msdr_fir_%s_process" % (ch, n, nt, "q15" if q15 else "f32"),
                      "kernel": trn},
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "traffic": None, "kernel_ms": round(k_ms, 4), "launches_timed": int(launches),
                        "note": "%d B per sample; kernel_ms from HIP events around the stage's main kernel (its history kernel, ~6 us, is in ms_per_step only)" % int(bps)}}
    attach_traffic(out, "fir_q15" if q15 else "fir_f32", args)
    if power:
        out["roofline"].update(power)
    if not q15 and trn.startswith("fir_f32tq_kernel<"):
        # 4 sub-tiles x 3 products x NS k-steps of v_mfma_f32_16x16x32_f16 (16384 flop each) per 1024-output tile
        ns_steps = int(trn.split("<")[1].split(",")[0].split(">")[0])
        tf = 12.0 * ns_steps * 16384.0 / 1024.0 * ch * n / (k_ms * 1e-3) / 1e12
        out["roofline"]["mfma_f16_tflops_executed"] = round(tf, 1)
        out["roofline"]["mfma_f16_frac"] = round(tf / MFMA_F16_PEAK_TFLOPS, 4)
        say_which_roof(out["roofline"], bps)
    if do_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orclib
        orc = orclib.Oracle()
        t1 = time.perf_counter()
        worst, nsamp = 0.0, 0
        for (r, lo), (xs, got) in keep.items():
            skip = xs.size - got.size                            # pre-roll samples in front of the window
            m = (xs.size // 128) * 128
            if q15:
                _, want = orc.fir_q15_blocks(taps, xs[:m], 128)
                want = want[skip:]
                worst = max(worst, float((want != got[:want.size]).sum()))
            else:
                want = orc.fir_f32_blocks(taps, xs[:m], 128)[skip:].astype(np.float64)
                worst = max(worst, float(np.sqrt(((want - got[:want.size]) ** 2).sum() / max((want ** 2).sum(), 1e-300))))
            nsamp += xs.size
        # the timed CPU leg: the oracle's FIR over rows of the same input, an OpenMP team sized like the headline's (pick_team)
        xs_all = np.ascontiguousarray(x_cpu if not q15 else x_cpu[:, :(x_cpu.shape[1] // 128) * 128])
        ys_all = np.empty_like(xs_all)
        tp = np.ascontiguousarray(taps)
        fn_b = orc.lib.orc_fir_q15_batch if q15 else orc.lib.orc_fir_f32_batch
        fn_b.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int]

        def run(xs_, modes_, ts_, thr):
            xs_ = np.ascontiguousarray(xs_)
            t0 = time.perf_counter()
            used = fn_b(tp.ctypes.data, C.c_uint16(nt), xs_.ctypes.data, ys_all.ctypes.data, C.c_uint32(xs_.shape[0]), C.c_uint64(xs_.shape[1]), C.c_uint32(128), int(thr))
            return None, time.perf_counter() - t0, used
        dummy = np.zeros(xs_all.shape[0], np.int32)
        cal = xs_all[:1, :min(xs_all.shape[1], 1 << 17)]
        _, dt1, _ = run(cal, dummy[:1], dummy[:1], 1)
        team, ladder = pick_team(run, xs_all, dummy, dummy, usable_cores()[0])
        total_dt, passes, used = 0.0, 0, 1
        while total_dt < 8.0 and passes < 5000:
            _, dtp, used = run(xs_all, dummy, dummy, team)
            total_dt += dtp
            passes += 1
        cores = int(min(used, xs_all.shape[0]))
        rate, rate1 = xs_all.size * passes / total_dt, cal.size / dt1
        out["cpu_baseline"] = dict({"value": round(rate / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                    "one_thread_Msamples_per_s": round(rate1 / 1e6, 3), "scaling_vs_one_thread": round(rate / (cores * rate1), 3),
                                    "team_ladder_Msamples_per_s": ladder,
                                    "sample": "%d rows x %d samples of the same input, %d pass(es), %.1f s wall (oracle/msdr_oracle.c orc_fir_%s_batch: %s in "
                                              "128-sample calls; OpenMP team of %d = the best rate on this lease)"
                                              % (xs_all.shape[0], xs_all.shape[1], passes, total_dt, "q15" if q15 else "f32", "arm_fir_fast_q15" if q15 else "arm_fir_f32", cores)},
                                   **host_info())
        out["parity"] = {("mismatching_samples" if q15 else "rel_rms_worst"): float("%.3g" % worst), "tolerance": 0 if q15 else 1e-6,
                         "windows": [{"channel": int(r), "start": int(lo), "length": int(L)} for (r, lo) in keep]}
    ceiling_fracs(out["roofline"])
    return out


def profile_kernel_name(path):
    """The library kernel a committed profile summary was taken on: its `dominant kernel:` line (tools/prof_summary.py, round 3 on)
    or, in older summaries, the first msdr:: row of the kernel-stats table.  Returns the name without template arguments."""
    # the kernel the COUNTER passes were taken on, where the summary names it (the row under "FETCH_SIZE counter:"): in a record whose
    # step runs other kernels behind the chain kernel (q15_c3: the biquad nodes take longer than the demodulator) the kernel with the most
    # time is not the one the traffic figure belongs to
    prev = ""
    for line in open(path):
        if prev.startswith("FETCH_SIZE counter:") and "msdr::" in line:
            return line.split("msdr::", 1)[1].split("<")[0].split("(")[0].split()[0]
        prev = line
    for line in open(path):
        if line.startswith("dominant kernel:"):
            return line.split(":", 1)[1].strip().split("<")[0].replace("void ", "").replace("msdr::", "")
    for line in open(path):
        if "msdr::" in line and "calls" in line:
            return line.split("msdr::", 1)[1].split("<")[0].split("(")[0].strip()
    return None


def attach_traffic(out, tag, args):
    """HBM traffic per launch of the dominant kernel.  The PMC passes cannot run inside this process (rocprofv3 wraps the command),
    so the figure is the one tools/profile.sh measured for this workload and committed under profiles/ (FETCH_SIZE x 1024 x 2 +
    WRITE_SIZE x 1024, corrected as MI355X_MICROARCH.md prescribes).  It is attached only when the profile was taken on the kernel
    THIS run launched (names compared) and the run is the named configuration: any shape override, experiment switch or MSDR_*
    kernel-selection variable leaves `traffic` null."""
    named = getattr(args, "named_record", None)          # a sub-record of the default run whose shape override IS its definition (fir512, c3_b128, c3_i16 ...): its own tag
    if named:
        tag = named
    elif (args.samples or args.channels or args.taps or args.stages >= 0 or args.no_mfma or args.no_fold or getattr(args, "out_i16", False)
            or args.time_segments or args.osc_period != 4):
        return
    if any(k.startswith("MSDR_") and k not in ("MSDR_LIB", "MSDR_BENCH_REHEARSAL", "MSDR_BENCH_NO_POWER") for k in os.environ):
        return
    ran = str(out["config"].get("kernel", "")).split("<")[0].split(" ")[0]
    for rnd in ("r05", "r04", "r03", "r02", "r01"):
        prof = os.path.join(ROOT, "profiles", rnd, "%s_rocprof_summary.txt" % tag)
        if not os.path.exists(prof):
            continue
        pk = profile_kernel_name(prof)
        if pk is None or pk != ran:
            out["roofline"]["traffic_note"] = "profiles/%s/%s_rocprof_summary.txt was taken on %s, this run launched %s: not attached" % (rnd, tag, pk, ran)
            continue
        for line in open(prof):
            if line.startswith("HBM traffic per chain_kernel launch") and "total" in line:
                out["roofline"]["traffic"] = float(line.rsplit("total", 1)[1].split()[0])
                out["roofline"]["traffic_source"] = "profiles/%s/%s_rocprof_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on %s)" % (rnd, tag, pk)
        if out["roofline"]["traffic"] is not None:
            out["roofline"].pop("traffic_note", None)
            return


def parity_windows(wl, info, x, y, q15, torch):
    """(lo, length, channel) windows of the GPU output to check against the oracle: the head, stretches that straddle boundaries
    between the kernel's time segments (first, middle, last boundary), and the tail; on the first, a middle and the last channel."""
    ch, n = wl["channels"], wl["n"]
    L = min(n, 4096)
    nseg = max(1, int(info["time_segments"]))
    tile = max(1, int(info["tile"]))
    seg = -(-n // nseg)
    seg = -(-seg // tile) * tile
    los, bounds = [0], []
    if nseg > 1 and seg < n:
        last = (n - 1) // seg
        for k in sorted({1, max(1, last // 2), last}):
            if k * seg + L // 2 <= n and k * seg - L // 2 > 0:
                los.append(k * seg - L // 2)
                bounds.append(k * seg)
    los.append(n - L)
    chans = sorted({0, ch // 2, ch - 1})
    wins = [(c, lo) for c in chans for lo in sorted(set(los))]
    if q15 and len(wl["bq"]):            # the Teensy biquad nodes are nonlinear: their state cannot be re-created by a pre-roll
        wins = [(c, 0) for c in chans]
        bounds = []
    return wins, L, bounds


def chain_parity_capture(wl, info, x, y, q15, torch):
    """The parity windows' IF input and GPU audio, copied to the host right after the first pass (small); the oracle runs on them
    AFTER the timed region, so that the card does not sit idle (and fall to its idle clock) just before the warm-up steps."""
    wins, L, bounds = parity_windows(wl, info, x, y, q15, torch)
    period = 128 if wl["osc"] else 4
    preroll = 8192
    cap = []
    for c, lo in wins:
        start = max(0, lo - preroll)
        start -= start % period
        cap.append((c, lo, start, x[c, start:lo + L].cpu().numpy(), y[c, lo:lo + L].cpu().numpy()))
    return cap, L, bounds, preroll


def chain_parity(wl, captured, q15, orc, orclib, out_i16=False):
    """GPU output against the oracle on parity_windows(); the oracle runs over [lo - preroll, lo + L) restarted on an oscillator
    period boundary (FIR history and IIR state settle inside the pre-roll) -- tests/test_gpu_fullsize.py:_window_check."""
    cap, L, bounds, preroll = captured
    osc_i, osc_q = wl["osc"] if wl["osc"] else (np.array([0, 1, 0, -1], np.float32), np.array([1, 0, -1, 0], np.float32))
    worst, checked = 0.0, 0
    nodes = None
    for c, lo, start, xs, got in cap:
        mode = int(wl["modes"][c]) if wl["modes"] is not None else wl["mode"]
        ts = int(wl["tapsets"][c]) if wl["tapsets"] is not None else 0
        if q15:
            m = (xs.size // 128) * 128
            nodes = [orc.biquad_teensy_new(cf) for cf in wl["qnodes"]]
            oi, oq = wl["qosc"] if wl["qosc"] else (None, None)
            want = orc.chain_q15(xs[:m], mode, wl["qi"][ts], wl["qq"][ts], mixer=1 if oi is not None else 0, osc_i=oi, osc_q=oq,
                                 biquads=nodes)[lo - start:]
            worst = max(worst, float((want != got[:want.size]).sum()))
        elif out_i16:      # the oracle's fp32 audio converted as arm_float_to_q15 does (truncate toward zero, saturate): at most 1 LSB apart
            wf = orc.chain_f32(xs, mode, wl["ci"][ts], wl["cq"][ts], osc_i, osc_q, wl["bq"] if len(wl["bq"]) else None)[lo - start:]
            want = np.clip(np.trunc((wf * np.float32(32768.0)).astype(np.float64)), -32768, 32767).astype(np.int32)
            worst = max(worst, float(np.abs(want - got.astype(np.int32)).max()))
        else:
            want = orc.chain_f32(xs, mode, wl["ci"][ts], wl["cq"][ts], osc_i, osc_q, wl["bq"] if len(wl["bq"]) else None)[lo - start:].astype(np.float64)
            worst = max(worst, float(np.sqrt(((want - got) ** 2).sum() / max((want ** 2).sum(), 1e-300))))
        checked += L
    if out_i16 and not q15:
        return {"max_abs_lsb": float(worst), "tolerance": 1, "windows": [{"channel": int(c), "start": int(lo), "length": int(L)} for c, lo, _, _, _ in cap],
                "segment_boundaries_covered": [int(b) for b in bounds], "oracle_preroll": preroll, "samples_checked": int(checked)}
    return {("mismatching_samples" if q15 else "rel_rms_worst"): float("%.3g" % worst), "tolerance": 0 if q15 else 1e-5,
            "windows": [{"channel": int(c), "start": int(lo), "length": int(L)} for c, lo, _, _, _ in cap],
            "segment_boundaries_covered": [int(b) for b in bounds], "oracle_preroll": preroll, "samples_checked": int(checked)}


def bench_chain(args, name, torch, msdr, ctx, dev, rank, world, dist, do_cpu, do_gather):
    """One record of the fused chain (c2 .. c5)."""
    wl = workload(name, msdr, rank, args.osc_period, args.lowpass)
    if args.samples:
        wl["n"] = args.samples
    if args.channels:
        wl["channels"] = args.channels
        if wl["modes"] is not None:
            wl["modes"], wl["tapsets"] = np.resize(wl["modes"], args.channels), np.resize(wl["tapsets"], args.channels)
    if args.stages >= 0:
        wl["bq"] = wl["bq"][:args.stages]
        wl["name"] += " [experiment: %d biquad stages]" % args.stages
    if args.taps:
        pair = len(wl["ci"]) == 1 and wl["mode"] == msdr.MODE_LSB
        if pair:
            hi, hq = hilbert_pair(args.taps)
            wl["ci"], wl["cq"] = [hi], [hq]
        else:
            lp = lowpass(args.taps)
            hi, hq = hilbert_pair(args.taps)
            wl["ci"], wl["cq"] = ([lp], [lp]) if len(wl["ci"]) == 1 else ([lp, hi], [lp, hq])
        wl["taps"] = args.taps
        wl["name"] += " [experiment: %d taps]" % args.taps
    ch, n = wl["channels"], wl["n"]
    q15 = args.arith == "q15"
    i16 = bool(getattr(args, "out_i16", False)) and not q15     # fp32 chain, int16 audio out (MSDR_CHAIN_OUT_I16): 4 B per sample
    if q15:       # the same workload through the as-written integer chain: Q15 taps / oscillator, Teensy biquad nodes
        corr = msdr.AUDIO_SAMPLE_RATE_EXACT / FS
        wl["qi"] = [np.round(np.asarray(c, np.float64) * 32767).astype(np.int16) for c in wl["ci"]]
        wl["qq"] = [np.round(np.asarray(c, np.float64) * 32767).astype(np.int16) for c in wl["cq"]]
        wl["qosc"] = tuple(np.round(np.asarray(o, np.float64) * 32768).clip(-32768, 32767).astype(np.int16) for o in wl["osc"]) if wl["osc"] else None
        wl["qnodes"] = [[msdr.biquad_design(msdr.BQ_LOWPASS, np.float32(6000 * 0.9 * corr), 0.54)],
                        [msdr.biquad_design(msdr.BQ_NOTCH, np.float32(FS / 8 * corr), 15.0)]][:len(wl["bq"])]
        chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, wl["qi"], wl["qq"], mixer=wl["mixer"], mode=wl["mode"], modes=wl["modes"],
                           tapsets=wl["tapsets"], osc_i=wl["qosc"][0] if wl["qosc"] else None,
                           osc_q=wl["qosc"][1] if wl["qosc"] else None, biquad_nodes=wl["qnodes"])
    else:
        chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, wl["ci"], wl["cq"], mixer=wl["mixer"], mode=wl["mode"], modes=wl["modes"],
                           tapsets=wl["tapsets"], osc_i=wl["osc"][0] if wl["osc"] else None, osc_q=wl["osc"][1] if wl["osc"] else None,
                           biquad_coeffs=wl["bq"] if len(wl["bq"]) else None, time_segments=args.time_segments,
                           flags=(msdr.CHAIN_NO_TAP_FOLDING if args.no_fold else 0)
                           | (msdr.CHAIN_NO_MFMA if args.no_mfma else 0) | (msdr.CHAIN_OUT_I16 if i16 else 0))
    x = synth_if(torch, dev, ch, n, wl["seed"])
    y = torch.empty((ch, n), dtype=torch.int16 if (q15 or i16) else torch.float32, device=dev)
    torch.cuda.synchronize(dev)

    # first pass from zero state: kept for the parity checks (the timed passes continue the stream, state carried)
    chain.process(x.data_ptr(), y.data_ptr(), n)
    torch.cuda.synchronize(dev)
    info = chain.info()
    parity = None
    first_rows = min(ch, max(2 * usable_cores()[0], 16))        # IF rows kept for the CPU leg: at least two per thread of its team
    keep = min(n, 1 << 22)                                     # GPU audio kept for the head comparison with the timed CPU sample
    keep_x = min(n, (1 << 26) if ch == 1 else (1 << 22))       # IF sample handed to the CPU baseline
    gpu_first = None
    captured = None
    if rank == 0 and (do_cpu or args.parity):
        captured = chain_parity_capture(wl, info, x, y, q15, torch)          # small host copies; the oracle runs after the timed region
        if do_cpu:
            gpu_first = y[:first_rows, :keep].clone()                        # on the device for now
    if n <= 1024:
        # block cadence: a tick is ~10 us, and the two HIP events around the main kernel (created and recorded per call) cost a good part of
        # that on the host -- the K timed steps run WITHOUT them (ms_per_step / value = the tick as a caller sees it), the kernel's own time
        # comes from K more steps with the events on
        dt = timed_steps(args, torch, dev, dist, lambda: chain.process(x.data_ptr(), y.data_ptr(), n))
        # device time per tick: ONE event pair around K more back-to-back calls on the launch stream (kernel + the dispatch gap to the next
        # one; per-call event pairs cost ~3 us each at this size and would time themselves); rocprofv3's per-kernel average: profiles/r05
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            chain.process(x.data_ptr(), y.data_ptr(), n)
        e1.record()
        torch.cuda.synchronize(dev)
        kernel_ms, launches = e0.elapsed_time(e1), args.steps
        # the same ticks as replays of ONE HIP graph of eight calls (msdr_chain_graph_*): what a host loop pays per call goes away
        graph_tick_us = None
        if dist is None:
            try:
                g = chain.graph([x.data_ptr()] * 8, [y.data_ptr()] * 8, n)
                reps = max(1, args.steps // 8)
                for _ in range(max(4, reps // 4)):
                    g.launch()
                torch.cuda.synchronize(dev)
                tg = time.perf_counter()
                for _ in range(reps):
                    g.launch()
                torch.cuda.synchronize(dev)
                graph_tick_us = (time.perf_counter() - tg) / (reps * 8) * 1e6
                g.close()
            except msdr.MsdrError as e:
                graph_tick_us = None
                args.graph_note = str(e)[:160]
    else:
        chain.enable_timing(True)
        dt = timed_steps(args, torch, dev, dist, lambda: chain.process(x.data_ptr(), y.data_ptr(), n), after_warmup=chain.kernel_time)
        kernel_ms, launches = chain.kernel_time()
        chain.enable_timing(False)
    power = power_probe(lambda: chain.process(x.data_ptr(), y.data_ptr(), n), torch, dev, dev.index or 0, world) if rank == 0 else None
    if captured is not None:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orclib
        parity = chain_parity(wl, captured, q15, orclib.Oracle(), orclib, out_i16=i16)
        if gpu_first is not None:
            gpu_first = gpu_first.cpu().numpy()

    gather = None
    if dist is not None and do_gather:                         # RCCL gather of demodulated audio, timed on its own
        import msdr_dist
        esz = y.element_size()
        rows = max(1, min(ch, (1 << 24) // n)) if n <= (1 << 24) else 1
        cols = min(n, 1 << 24)
        part = y[:rows, :cols].contiguous().to(args.cdev)       # a bounded slice of this rank's audio shard
        m = rows * cols

        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize(dev)
            dist.barrier()
            g0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(dev)
            dist.barrier()
            return msdr_dist.max_over_ranks((time.perf_counter() - g0) / reps, args.cdev)
        t_all = timed(lambda: msdr_dist.gather_audio(part, world * rows))
        t_root = timed(lambda: msdr_dist.gather_audio_root(part, world * rows, root=0))
        # double-buffered schedule on a 128-sample-per-channel cadence is launch-bound; use the bench block: compute into buffer
        # k & 1, submit its gather, go on with block k + 1
        og = msdr_dist.OverlappedGather(world * rows, rows, cols, y.dtype, args.cdev, root=0)
        xs = x[:rows, :cols].contiguous()
        blocks = 6

        def pipeline(with_gather):
            for k in range(blocks):
                buf = og.buffer(k) if with_gather else og.bufs[k & 1]
                if args.cdev.type == "cuda":
                    chain_small.process(xs.data_ptr(), buf.data_ptr(), cols)
                else:                                        # rehearsal: the audio buffers live on the host
                    chain_small.process(xs.data_ptr(), ysmall.data_ptr(), cols)
                    buf.copy_(ysmall)
                if with_gather:
                    og.submit(k)
            if with_gather:
                og.finish()
        chain_small = make_chain(msdr, ctx, wl, rows, q15, args)
        ysmall = torch.empty((rows, cols), dtype=y.dtype, device=dev)
        t_comp = timed(lambda: pipeline(False), reps=2) / blocks
        t_ovl = timed(lambda: pipeline(True), reps=2) / blocks
        # the same from C (include/msdr.h msdr_comm_*: RCCL driven by the library, its own stream, event-ordered behind the compute)
        c_gather = None
        if args.cdev.type == "cuda":
            try:
                uid = [msdr.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                comm = msdr.Comm(ctx, uid[0], rank, world)
                recv = [torch.empty((world * rows, cols), dtype=y.dtype, device=dev) if rank == 0 else None for _ in range(2)]
                ybuf = [torch.empty((rows, cols), dtype=y.dtype, device=dev) for _ in range(2)]

                def c_root():
                    comm.begin(0, part.data_ptr(), m * esz, recv[0].data_ptr() if rank == 0 else None, 0)
                    comm.wait(0, host_wait=True)

                def c_pipeline():
                    for k in range(blocks):
                        comm.wait(k & 1)                       # device-side: the gather that last read this buffer
                        chain_small.process(xs.data_ptr(), ybuf[k & 1].data_ptr(), cols)
                        comm.begin(k & 1, ybuf[k & 1].data_ptr(), m * esz, recv[k & 1].data_ptr() if rank == 0 else None, 0)
                    comm.wait(0, host_wait=True)
                    comm.wait(1, host_wait=True)
                t_c = timed(c_root)
                t_cp = timed(c_pipeline, reps=2) / blocks
                ok = True
                if rank == 0:
                    ok = bool(torch.equal(recv[0][:rows], ybuf[0])) if blocks % 2 == 0 else True
                comm.close()
                c_gather = {"op": "msdr_gather_audio_begin / _wait (RCCL send / recv group to rank 0, library-owned stream)",
                            "ms": round(t_c * 1e3, 3), "GBps_into_root": round((world - 1) * m * esz / t_c / 1e9, 1),
                            "ms_per_block_with_gather_overlapped": round(t_cp * 1e3, 3),
                            "gather_time_hidden_frac": round(max(0.0, min(1.0, 1.0 - (t_cp - t_comp) / max(t_c, 1e-9))), 3), "root_received_own_shard": ok}
            except Exception as e:                           # the torch.distributed figures above stand on their own
                c_gather = {"error": str(e)[:200]}
        chain_small.close()
        gather = {"audio_dtype": str(y.dtype).replace("torch.", ""), "bytes_per_rank": m * esz, "ranks": world,
                  "rank_devices": args.rank_devices,
                  "all_gather": {"op": "msdr_dist.gather_audio = all_gather_into_tensor", "ms": round(t_all * 1e3, 3),
                                 "GBps_into_each_rank": round((world - 1) * m * esz / t_all / 1e9, 1), "Msamples_per_s": round(world * m / t_all / 1e6, 1)},
                  "gather_to_root": {"op": "msdr_dist.gather_audio_root = gather(dst=0)", "ms": round(t_root * 1e3, 3),
                                     "GBps_into_root": round((world - 1) * m * esz / t_root / 1e9, 1), "Msamples_per_s": round(world * m / t_root / 1e6, 1)},
                  "overlapped": {"op": "msdr_dist.OverlappedGather: gather(k) to root while block k + 1 is demodulated, two audio buffers",
                                 "block": "%d channels x %d samples per rank" % (rows, cols), "ms_per_block_compute_only": round(t_comp * 1e3, 3),
                                 "ms_per_block_with_gather": round(t_ovl * 1e3, 3),
                                 "gather_time_hidden_frac": round(max(0.0, min(1.0, 1.0 - (t_ovl - t_comp) / max(t_root, 1e-9))), 3)},
                  "c_abi": c_gather}
    x_host = x[:first_rows, :keep_x].cpu().numpy() if (rank == 0 and do_cpu) else None
    chain.close()
    del x, y
    torch.cuda.empty_cache()
    if rank != 0:
        return None

    samples_per_step = ch * n
    value = world * samples_per_step * args.steps / dt / 1e6
    k_ms = kernel_ms / max(launches, 1)
    alg_bytes = (4.0 if (q15 or i16) else 6.0) * samples_per_step        # int16 in + fp32 (or int16) out
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    extra = (6 if wl["mixer"] == msdr.MIXER_NCO else 0) + 4 + 9 * len(wl["bq"])
    flop_written = 4.0 * wl["taps"] + extra                      # as the reference writes it: two N-tap FIRs
    folded = info["kernel"].startswith("chain_fold")
    flop_exec = (2.0 if folded else 4.0) * info["taps_padded"] + extra     # what the kernel executes (tap folding halves the MACs)
    out = {
        "metric": "Msamples/s through IF->I/Q->FIR->demod->IIR chain; achieved HBM GB/s vs peak",
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "q15 (int16 data, wrapping int32 accumulate)" if q15 else "f32 (fp32 accumulate; matrix-core operands as 2 x fp16 pieces: int16 samples exact, taps 22 bits)",
        "data": "synthetic",
        "config": {"workload": wl["name"], "channels_per_gpu": ch, "samples_per_channel_per_step": n, "taps": wl["taps"],
                   "biquad_stages": int(len(wl["bq"])), "in": "int16", "out": "int16" if (q15 or i16) else "fp32", "sharding": "independent channels per GPU, no data-path collective",
                   "kernel": info["kernel"], "grid": info["grid"], "time_segments": info["time_segments"], "iir_warmup": info["warmup"],
                   "tap_folding": not args.no_fold},
        # (four decimals, or -- a single receiver moves 47 MB/s -- two significant digits)
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1) if achieved >= 1.0 else float("%.2g" % achieved), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved / HBM_PEAK_GBS >= 1e-3 else float("%.2g" % (achieved / HBM_PEAK_GBS)), "traffic": None,
                     "kernel_ms": round(k_ms, 4), "launches_timed": int(launches),
                     "valu_tflops_executed": round(flop_exec * samples_per_step / (k_ms * 1e-3) / 1e12, 2),
                     "valu_frac_executed": round(flop_exec * samples_per_step / (k_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4),
                     "valu_peak_tflops": VALU_PEAK_TFLOPS,
                     "as_written_equivalent_tflops": round(flop_written * samples_per_step / (k_ms * 1e-3) / 1e12, 2)},
    }
    if info["kernel"].startswith("chain_mf") or info["kernel"].startswith("chain_tr"):
        # matrix-core kernel: the folded FIR runs as 3 fp16 matrix products (32768 flop each in 32x32x16 terms) per k-step and
        # 1024-output wave tile; the vector ALU only carries staging, demod and the IIR scan, so the valu_* fields do not apply
        r = out["roofline"]
        for k in ("valu_tflops_executed", "valu_frac_executed", "valu_peak_tflops"):
            r.pop(k)
        mf = 3 * 32768.0 * info["mfma_ksteps"] / 1024.0
        r["mfma_f16_tflops_executed"] = round(mf * samples_per_step / (k_ms * 1e-3) / 1e12, 1)
        r["mfma_f16_frac"] = round(mf * samples_per_step / (k_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)
        r["mfma_f16_peak_tflops"] = MFMA_F16_PEAK_TFLOPS
    if info["kernel"].startswith("chain_q15mf"):
        # integer matrix-core kernel: four v_mfma_i32_32x32x32_i8 (65536 integer ops each) per k-step and 1024-output wave tile
        r = out["roofline"]
        for k in ("valu_tflops_executed", "valu_frac_executed", "valu_peak_tflops"):
            r.pop(k, None)
        mi = 4 * 65536.0 * info["mfma_ksteps"] / 1024.0
        r["mfma_i8_tops_executed"] = round(mi * samples_per_step / (k_ms * 1e-3) / 1e12, 1)
        r["mfma_i8_frac"] = round(mi * samples_per_step / (k_ms * 1e-3) / 1e12 / (2 * MFMA_F16_PEAK_TFLOPS), 4)
        r["mfma_i8_peak_tops"] = 2 * MFMA_F16_PEAK_TFLOPS
        r["note"] = "kernel_ms / achieved are the FIR + demod kernel (4 B/sample); the step also runs the Teensy biquad nodes (serial per channel)"
    attach_traffic(out, ("q15_" if q15 else "") + name, args)
    if power:
        out["roofline"].update(power)
    if n <= 1024:
        out["roofline"]["tick_us"] = round(dt / args.steps * 1e6, 2)
        out["roofline"]["graph_tick_us"] = round(graph_tick_us, 2) if graph_tick_us else None
        if graph_tick_us:
            out["roofline"]["graph_Msamples_per_s"] = round(samples_per_step / graph_tick_us, 1)
        out["roofline"]["note_block"] = "block cadence: kernel_ms = device time per call from one HIP event pair around K back-to-back calls (kernels + dispatch gaps)"
    if q15 and len(wl["bq"]) and n > 1024:
        # the as-written step = the FIR + demod kernel, then the two Teensy biquad nodes in place on the audio (serial per channel)
        out["roofline"]["step_ms"] = round(dt / args.steps * 1e3, 4)
        out["roofline"]["demod_kernel_ms"] = round(k_ms, 4)
        out["roofline"]["node_pass_ms"] = round(dt / args.steps * 1e3 - k_ms, 4)
    if gather:
        out["gather"] = gather
    if do_cpu:
        cb, worst, per_row = cpu_baseline_q15(wl, x_host, gpu_first) if q15 else cpu_baseline(wl, x_host, gpu_first)
        out["cpu_baseline"] = dict(cb, **host_info())
        parity["head_vs_timed_cpu_sample"] = {("mismatching_samples" if q15 else "rel_rms_worst"): float("%.3g" % worst),
                                              "rows": int(min(first_rows, max(1, x_host.shape[0]))), "samples_per_row": int(per_row)}
    if parity is not None:
        out["parity"] = parity
    say_which_roof(out["roofline"], alg_bytes / samples_per_step)
    ceiling_fracs(out["roofline"])
    return out


def emit_line(out):
    """The ONE line rank 0 prints.  The full record (every sub-record with its parity windows, CPU ladder, gather figures) is written to
    bench_full.json beside this script; the line itself stays under ~6 KB so that a log tail keeps all of it: the sub-records travel as
    compact summaries INSIDE the top-level `roofline` (`roofline.records`), window lists and long notes are left to the file."""
    full_path = os.path.join(ROOT, "bench_full.json")
    try:
        with open(full_path, "w") as f:
            json.dump(out, f, indent=1)
        gdir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(gdir):
            with open(os.path.join(gdir, "bench_full.json"), "w") as f:
                json.dump(out, f, indent=1)
    except OSError:
        full_path = None
    line = {k: v for k, v in out.items() if k not in ("also", "gather", "parity", "cpu_baseline", "roofline", "config")}
    cfg = dict(out.get("config", {}))
    cfg.pop("records", None)
    line["config"] = cfg
    r = dict(out.get("roofline", {}))
    r.pop("power_samples", None)
    if "also" in out:
        r["records"] = {k: compact_record(v) for k, v in out["also"].items() if v is not None}
    line["roofline"] = r
    cb = out.get("cpu_baseline")
    if cb:
        cb = {k: v for k, v in cb.items() if k not in ("team_ladder_Msamples_per_s",)}
        if isinstance(cb.get("sample"), str):
            cb["sample"] = cb["sample"][:160]
        line["cpu_baseline"] = cb
    par = out.get("parity")
    if par:
        line["parity"] = {k: v for k, v in par.items() if k != "windows"}
    g = out.get("gather")
    if g:
        line["gather"] = {"ranks": g.get("ranks"), "audio_dtype": g.get("audio_dtype"), "bytes_per_rank": g.get("bytes_per_rank"),
                          "all_gather_ms": g.get("all_gather", {}).get("ms"), "gather_to_root_ms": g.get("gather_to_root", {}).get("ms"),
                          "gather_to_root_GBps": g.get("gather_to_root", {}).get("GBps_into_root"),
                          "overlapped_ms_per_block": g.get("overlapped", {}).get("ms_per_block_with_gather"),
                          "compute_only_ms_per_block": g.get("overlapped", {}).get("ms_per_block_compute_only"),
                          "hidden_frac": g.get("overlapped", {}).get("gather_time_hidden_frac")}
    line["full_record"] = "bench_full.json" if full_path else None
    if len(json.dumps(line)) > 6000:                       # belt and braces: the summaries first, then the long strings
        line["config"] = {k: (v[:80] if isinstance(v, str) else v) for k, v in line["config"].items()}
        if isinstance(line.get("dtype"), str):
            line["dtype"] = line["dtype"][:60]
    return line


def bench_update_all(args, msdr, rank):
    """The tick through the reference's own operator API: tests/cpp/bench_ticks (C++, the product library only) runs AudioStream::update_all()
    once per 128 samples over source -> AudioSDRDemodulator (Q15 demodulation() + biquad1_dac + biquad2_dac) -> sink, as a child process;
    its first output block is checked here against the oracle (bit-exact)."""
    if rank != 0:
        return None
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "bench_ticks")
    if not os.path.exists(exe):
        return None
    ch, ticks, taps = args.channels or 4096, max(args.steps, 200) * 10, 256
    try:
        r = subprocess.run([exe, str(ch), str(ticks), str(taps)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except (OSError, subprocess.SubprocessError, IndexError, ValueError) as e:
        return {"config": {"workload": "update_all", "kernel": "failed: %s" % str(e)[:100]}, "roofline": {"frac": 0.0, "kernel_ms": 0.0}, "parity": None,
                "value": 0.0, "ms_per_step": 0.0}
    # parity: the program's input is an LCG (bench_ticks.cpp) and the same block every tick; its first output block comes from zero state
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    orc = orclib.Oracle()
    s, vals = 12345, []
    for _ in range(4 * 128):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        vals.append((s >> 16) % 16001 - 8000)
    x = np.array(vals, np.int16).reshape(4, 128)
    corr = msdr.AUDIO_SAMPLE_RATE_EXACT / FS
    am = msdr.calc_fir_coeffs(taps, 2800.0, 70.0, 0, 0.0, FS)[:taps].copy()
    lp = orc.biquad_design(orclib.BQ_LOWPASS, np.float32(6000 * 0.9 * corr), 0.54)
    nt = orc.biquad_design(orclib.BQ_NOTCH, np.float32(FS / 8 * corr), 15.0)
    got = np.array(d.pop("first_block_ch0_3"), np.int16).reshape(4, 128)
    bad = 0
    for c in range(min(4, ch)):
        want = orc.chain_q15(x[c], orclib.AM, am, am, biquads=[orc.biquad_teensy_new([lp]), orc.biquad_teensy_new([nt])])
        bad += int((want != got[c]).sum())
    tick_ms = d["tick_us"] / 1e3
    gbs = 4.0 * ch * 128 / (tick_ms * 1e-3) / 1e9
    return {"value": d["Msamples_per_s"], "ms_per_step": round(tick_ms, 5), "dtype": "q15 (int16 data, wrapping int32 accumulate)",
            "config": {"workload": "update_all: %d AM channels x 128 per AudioStream::update_all() tick, %d-tap designer low-pass pair + biquad1_dac + biquad2_dac, %d ticks" % (ch, taps, d["ticks"]),
                       "kernel": "update_all: d2d copy + chain_q15mb_kernel with both biquad nodes as its second phase", "graph": d["graph"], "steps_timed": d["ticks"]},
            "roofline": {"bound": "latency (two launches per tick, the node recursion serial per channel)", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": None, "kernel_ms": round(tick_ms, 5), "tick_us": round(d["tick_us"], 2),
                         "note_block": "wall time per update_all() tick of the C++ graph runtime, ticks queued back to back (kernel_ms = the same figure: not separated here)"},
            "parity": {"mismatching_samples": float(bad), "tolerance": 0, "samples_checked": int(min(4, ch) * 128),
                       "windows": [{"channel": c, "start": 0, "length": 128} for c in range(min(4, ch))]}}


def make_chain(msdr, ctx, wl, channels, q15, args):
    """The workload's chain for `channels` channels (the gather pipeline runs it on a bounded slice of the shard)."""
    modes = wl["modes"][:channels] if wl["modes"] is not None else None
    tapsets = wl["tapsets"][:channels] if wl["tapsets"] is not None else None
    if q15:
        return msdr.Chain(ctx, msdr.ARITH_Q15, channels, wl["qi"], wl["qq"], mixer=wl["mixer"], mode=wl["mode"], modes=modes, tapsets=tapsets,
                          osc_i=wl["qosc"][0] if wl["qosc"] else None, osc_q=wl["qosc"][1] if wl["qosc"] else None, biquad_nodes=wl["qnodes"])
    return msdr.Chain(ctx, msdr.ARITH_F32, channels, wl["ci"], wl["cq"], mixer=wl["mixer"], mode=wl["mode"], modes=modes, tapsets=tapsets,
                      osc_i=wl["osc"][0] if wl["osc"] else None, osc_q=wl["osc"][1] if wl["osc"] else None,
                      biquad_coeffs=wl["bq"] if len(wl["bq"]) else None)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """Runs this script as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same arguments>` in a child
    process and returns its exit code; the child's stdout (rank 0's one JSON line) and stderr pass straight through."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="all", choices=["all", "c2", "c3", "c4", "c5", "fe", "spec", "fir"],
                    help="all (default) = headline c3 (BASELINE.json configs[2]) with fir / c2 / c4 / c5 as sub-records; c2..c5 = configs[1..4] alone; "
                         "fir = the FIR stage alone; fe = the front end (SURVEY 8 f1); spec = the spectrum FFT (f4)")
    ap.add_argument("--samples", type=int, default=0, help="override samples per channel per step")
    ap.add_argument("--channels", type=int, default=0, help="override channels per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg (the windowed parity check stays unless --no-parity)")
    ap.add_argument("--no-parity", dest="parity", action="store_false", help="skip the in-run parity check too")
    ap.add_argument("--no-fold", action="store_true", help="keep mixer and FIR as separate arithmetic steps")
    ap.add_argument("--no-mfma", action="store_true", help="keep the folded FIR on the fp32 VALU (no split-fp16 matrix-core kernel)")
    ap.add_argument("--time-segments", type=int, default=0)
    ap.add_argument("--arith", default="f32", choices=["f32", "q15"],
                    help="f32 = the north-star flavour (default); q15 = the reference as written (int16 out, bit-exact)")
    ap.add_argument("--out-i16", action="store_true", help="f32 arithmetic, int16 audio out (MSDR_CHAIN_OUT_I16, arm_float_to_q15): 4 B per sample instead of 6")
    ap.add_argument("--osc-period", type=int, default=4, help="experiment: NCO period in samples (4 = fs/4, the named config)")
    ap.add_argument("--stages", type=int, default=-1, help="experiment: override the number of biquad stages (0..2)")
    ap.add_argument("--taps", type=int, default=0, help="experiment: override the tap count (same designer)")
    ap.add_argument("--lowpass", default="numpy", choices=["designer", "numpy"],
                    help="the AM low-pass of c3 / c5: designer = the reference's calc_FIR_coeffs (linear phase about an integer index); numpy = a windowed sinc "
                         "centred on a half sample (rounds 1-3)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as a CHILD torch.distributed.run job (one process per GPU),
        # relay its output and exit with its code.  Nothing in this process has touched the GPU yet (torch is not even
        # imported), and the launcher is a child, never an exec of this process.
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    import msdr

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (a launcher started another rank count than --gpus says)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the library has no CPU path")
    # MSDR_BENCH_REHEARSAL=1: a dry run of the N>1 control flow on a ONE-GPU box -- every rank on device 0, gloo instead of
    # RCCL (RCCL refuses two ranks on one device), collectives on host tensors.  Never a measurement; the line says so.
    rehearsal = world > 1 and os.environ.get("MSDR_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    args.rank_devices = ["cuda:%d" % local_rank]
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        names = [None] * world
        dist.all_gather_object(names, "rank %d: cuda:%d (%s)" % (rank, local_rank, torch.cuda.get_device_name(dev)))
        args.rank_devices = names
        assert dist.get_world_size() == world
    args.cdev = torch.device("cpu") if rehearsal else dev      # where the collectives' tensors live

    # One explicit HIP stream shared by torch (input synthesis, events, RCCL ordering) and the library.
    # (torch's default stream is the NULL stream, which msdr_ctx_create reads as "create your own".)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = msdr.Context(local_rank, stream.cuda_stream)
    do_cpu = not args.no_cpu

    if args.workload == "fe":
        out = bench_frontend(args, torch, msdr, ctx, dev, rank, world, dist)
    elif args.workload == "spec":
        out = bench_spectrum(args, torch, msdr, ctx, dev, rank, world, dist)
    elif args.workload == "fir":
        args.min_warm_s = 0.2
        out = bench_fir_stage(args, torch, msdr, ctx, dev, rank, world, dist, do_cpu)
        if out is not None:
            out["warmup_steps_run"] = args.warmup_steps_run
    elif args.workload != "all":
        args.min_warm_s = 0.0 if args.workload == "c3" else 0.2          # the headline config keeps the contract's W steps exactly
        out = bench_chain(args, args.workload, torch, msdr, ctx, dev, rank, world, dist, do_cpu, True)
        if out is not None:
            out["warmup_steps_run"] = args.warmup_steps_run
    else:
        out = bench_chain(args, "c3", torch, msdr, ctx, dev, rank, world, dist, do_cpu, True)      # the headline, with the CPU baseline
        also = {}
        args.min_warm_s = 0.2                    # sub-records: warm up by time as well as by count (see timed_steps)
        if args.arith == "f32":
            also["fir"] = bench_fir_stage(args, torch, msdr, ctx, dev, rank, world, dist, do_cpu)
            if also["fir"] is not None:                  # (records exist on rank 0 only)
                also["fir"]["warmup_steps_run"] = args.warmup_steps_run
        if args.arith == "f32":
            # the same stage at C5's 512 taps (past the taps-in-registers layout's ~290: fir_f32mf_kernel, taps in LDS)
            keep_taps = args.taps
            args.taps, args.named_record = 512, "fir_f32_512"
            also["fir512"] = bench_fir_stage(args, torch, msdr, ctx, dev, rank, world, dist, do_cpu)
            args.taps, args.named_record = keep_taps, None
            if also["fir512"] is not None:
                also["fir512"]["warmup_steps_run"] = args.warmup_steps_run
        for name in ("c2", "c4", "c5"):
            also[name] = bench_chain(args, name, torch, msdr, ctx, dev, rank, world, dist, False, False)   # parity windows, no timed CPU leg
            if also[name] is not None:
                also[name]["warmup_steps_run"] = args.warmup_steps_run
        if args.arith == "f32":
            # the reference AS WRITTEN on the headline shape: arm_fir_fast_q15 pair + integer demod (bit-exact, i8 matrix cores), then
            # biquad1_dac / biquad2_dac (Teensy Q2.30 biquads with error feedback: one lane per channel, latency-bound by construction)
            args.arith = "q15"
            also["q15_c3"] = bench_chain(args, "c3", torch, msdr, ctx, dev, rank, world, dist, False, False)
            args.arith = "f32"
            # the headline with int16 audio out (the play queue's type): the same arithmetic, 4 B per sample through HBM instead of 6
            args.out_i16, args.named_record = True, "c3_i16"
            also["c3_i16"] = bench_chain(args, "c3", torch, msdr, ctx, dev, rank, world, dist, False, False)
            args.out_i16, args.named_record = False, None
            if also["c3_i16"] is not None:
                also["c3_i16"]["warmup_steps_run"] = args.warmup_steps_run
            if also["q15_c3"] is not None:
                also["q15_c3"]["warmup_steps_run"] = args.warmup_steps_run
            # the reference's own cadence: ONE 128-sample AudioStream block per call (Minimal-SDR.ino:518-530, :574-575), c3's and c4's channel
            # counts; `tick_us` = wall time per call with the calls queued back to back.  200 timed steps at least (a tick is ~10-20 us).
            keep_steps, keep_samples = args.steps, args.samples
            args.steps, args.samples = max(args.steps, 200), 128
            # (q15_c1_b128: the reference's OWN size -- one receiver -- as the reference writes it: Q15, both biquad nodes)
            keep_channels = args.channels
            for name, wl_name, q, chs in (("c3_b128", "c3", False, 0), ("c4_b128", "c4", False, 0), ("q15_c3_b128", "c3", True, 0), ("q15_c1_b128", "c3", True, 1)):
                args.arith, args.named_record, args.channels = ("q15" if q else "f32"), name, (chs or keep_channels)
                also[name] = bench_chain(args, wl_name, torch, msdr, ctx, dev, rank, world, dist, False, False)
                args.named_record, args.channels = None, keep_channels
                if also[name] is not None:
                    also[name]["warmup_steps_run"] = args.warmup_steps_run
                    also[name]["config"]["steps_timed"] = args.steps
                    also[name]["roofline"]["tick_us"] = round(also[name]["ms_per_step"] * 1e3, 2)
            args.arith, args.steps, args.samples = "f32", keep_steps, keep_samples
            also["update_all"] = bench_update_all(args, msdr, rank)      # the same cadence through AudioStream::update_all() (C++ graph runtime, Q15)
        args.min_warm_s = 0.0
        if rank == 0:
            for k, rec in also.items():
                for drop in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "data"):
                    rec.pop(drop, None)
            out["also"] = also
            out["config"]["records"] = "headline = c3 (BASELINE.json configs[2]); also: fir (256-tap fp32 FIR stage alone), c2, c4, c5, q15_c3 (c3 through the reference's own integer arithmetic, bit-exact), c3_i16, and the 128-sample block cadence c3_b128 / c4_b128 / q15_c3_b128 / q15_c1_b128 (ONE receiver) -- each timed over the same K steps; their warm-up is W steps plus 0.2 s of untimed steps (warmup_steps_run), so that short steps are not timed on a card still climbing from its idle clock"
    if rank == 0:
        import ctypes as C
        ctx.lib.msdr_build_rev.restype = C.c_char_p
        out["library_rev"] = ctx.lib.msdr_build_rev().decode()       # the source revision lib/libmsdr.so was built from
        out["n_ranks_seen"] = dist.get_world_size() if dist is not None else 1
        out["rank_devices"] = args.rank_devices
        if rehearsal:
            out["rehearsal"] = "all %d ranks shared ONE GPU, gloo collectives on host tensors: control-flow dry run, not a measurement" % world
        print(json.dumps(emit_line(out)))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
