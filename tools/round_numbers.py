#!/usr/bin/env python3
"""The figures the documents quote from one round's evidence (profiles/<tag>_*): bench lines, kernel times, traffic, single-instance ticks.
usage: python tools/round_numbers.py <tag, e.g. r05_f>"""
import glob, json, os, re, sys
tag = sys.argv[1]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
for f in sorted(glob.glob(os.path.join(P, tag + "_bench*_plain.json"))) + sorted(glob.glob(os.path.join(P, tag + "*_bench.json"))):
    b = json.load(open(f)); r = b["roofline"]
    mm = b.get("ms_per_step_min_max") or [0, 0]
    print("%-52s %9.0f solves/s  %.2f ms/step (median %.2f, min/max %.1f/%.1f)  kernels %s  traffic %s GB  %s TB/s" % (
        os.path.basename(f), b["value"], b["ms_per_step"], b.get("ms_per_step_median") or 0, mm[0], mm[1],
        {k: round(v, 2) for k, v in r.get("kernel_ms", {}).items()},
        None if r.get("traffic") is None else round(r["traffic"] / 1e9, 1), None if r.get("traffic_GBs") is None else round(r["traffic_GBs"] / 1e3, 2)))
b = json.load(open(os.path.join(P, tag + "_bench_plain.json")))
print("cpu_baseline:", {k: v for k, v in b["cpu_baseline"].items() if k != "sample"})
for name in ("_latency_probe.txt", "_latency_probe_take2.txt"):
    f = os.path.join(P, tag + name)
    if os.path.exists(f):
        print("--", os.path.basename(f))
        for l in open(f):
            c = l.split("|")
            print("  " + " | ".join(x.strip() for x in (c[:1] + c[1:2] + c[3:5] + c[6:7])) if len(c) > 5 else "  " + l.strip())
f = os.path.join(P, tag + "_policy_audit.txt")
if os.path.exists(f):
    r = [float(m.group(1)) for m in (re.search(r"= ([0-9.]+) x the best", l) for l in open(f)) if m]
    print("policy audit: %d cells, default above 1.05 x the best column in %d, worst %.2f" % (len(r), sum(v > 1.05 for v in r), max(r)))
