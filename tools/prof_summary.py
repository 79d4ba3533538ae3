#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel time (rocprofv3 --stats) and per-launch HBM
traffic of the chain kernel from the FETCH_SIZE / WRITE_SIZE counter passes, corrected as
/opt/skills/guides prescribe: counters are in KiB (x1024), and on gfx950 FETCH_SIZE reports half of the
bytes of a wide (16 B/lane) coalesced streaming read (x2)."""
import csv
import glob
import json
import os
import sys

out, args = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
print("command: python3 bench.py", args)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("\n== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    lib = [r for r in rows if "msdr::" in r.get("Name", "")]
    if lib:
        print("dominant kernel:", lib[0]["Name"].split("(")[0])      # bench.py attach_traffic compares it with the run's config.kernel
    for r in rows[:12]:
        print("  %-70s calls %6s  total %12s ns  avg %12s ns  %5s %%" % (r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
line = [l for l in open(os.path.join(out, "trace.log")) if l.startswith("{")]
if line:
    j = json.loads(line[-1])
    print("\nbench line under the profiler: value %.1f %s, kernel_ms %s, achieved %.1f GB/s (frac %.4f)" % (
        j["value"], j["unit"], j["roofline"].get("kernel_ms", j.get("ms_per_step")), j["roofline"]["achieved"], j["roofline"]["frac"]))
    cfgj = j["config"]
    if "channels_per_gpu" in cfgj:
        q15 = cfgj.get("out") == "int16"
        alg = (4.0 if q15 else 6.0) * cfgj["channels_per_gpu"] * cfgj["samples_per_channel_per_step"]
        print("algorithmic bytes per launch: %.0f (2 B in + %d B out per sample)" % (alg, 2 if q15 else 4))
    elif cfgj["workload"].startswith("fir:"):     # FIR stage: "fir: C channels x N samples, ..."
        w = cfgj["workload"].replace(",", " ").split()
        bps = 4.0 if j.get("dtype") == "q15" else 8.0
        alg = bps * int(w[1]) * int(w[4])
        print("algorithmic bytes per launch: %.0f (%d B per sample)" % (alg, bps))
    else:                       # spectrum workload: 896 B per transform
        alg = 896.0 * int(cfgj["workload"].split()[1])
        print("algorithmic bytes per launch: %.0f (896 B per transform)" % alg)
res = {}
for name in ("fetch", "write"):
    # a step may launch the chain kernel more than once (c5: envelope channels and SSB channels are two instantiations): the figure per
    # STEP is the sum over the distinct library kernels of their mean per launch
    per = {}
    for f in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r.get("Kernel_Name", "")
            if ("chain_" in kn and "kernel" in kn) or "spectrum_rfft128" in kn or "fir_f32" in kn:
                a = per.setdefault(kn.split("(")[0], [0.0, 0])
                a[0] += float(r.get("Counter_Value", 0)); a[1] += 1
    tot = sum(a[0] / a[1] for a in per.values())
    cnt = sum(a[1] for a in per.values())
    res[name] = (tot, cnt, len(per))
    print("%s counter: %d chain_kernel dispatches of %d kernel(s), sum of the per-launch means %.1f" % (name.upper() + "_SIZE", cnt, len(per), tot))
    for kn, a in sorted(per.items()):
        print("    %-90s mean raw value %.1f over %d launches" % (kn[:90], a[0] / a[1], a[1]))
if res["fetch"][1] and res["write"][1]:
    fb = res["fetch"][0] * 1024 * 2      # KiB -> B, gfx950 wide-read correction x2
    wb = res["write"][0] * 1024
    print("HBM traffic per chain_kernel launch: read %.0f B (FETCH_SIZE x1024 x2), write %.0f B (WRITE_SIZE x1024), total %.0f B%s" % (
        fb, wb, fb + wb, "   [per STEP: the %d kernels of one step summed]" % res["fetch"][2] if res["fetch"][2] > 1 else ""))
    if line:
        print("traffic / algorithmic = %.3f" % ((fb + wb) / alg))
