#!/usr/bin/env python3
"""Fold a rocprofv3 `--kernel-trace --stats` kernel_stats.csv into kernel families (all template instantiations of the NT / TN
GEMM, LayerNorm, attention, ...): calls, total and average duration.
usage: rocprof_family_summary.py <kernel_stats.csv> [<kernel_trace.csv> <steps>]
The first step of a process also contains the GEMM variant measurements (a few launches of each candidate), which the --stats table cannot tell
from the timed steps.  With the per-dispatch trace and the number of steps the command ran (warm-up + timed), "last_steps" repeats the two GEMM
families over the LAST `steps` steps only (99 NT + 50 TN launches per ViT-B/16 step): the figure bench.py's own event timing must agree with."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import family  # noqa: E402

agg = collections.OrderedDict()
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    f = family(r["Name"])
    a = agg.setdefault(f, {"calls": 0, "total_ms": 0.0})
    a["calls"] += int(r["Calls"])
    a["total_ms"] += float(r["TotalDurationNs"]) / 1e6
out = collections.OrderedDict()
for f, a in sorted(agg.items(), key=lambda kv: -kv[1]["total_ms"]):
    out[f] = {"calls": a["calls"], "total_ms": round(a["total_ms"], 3), "avg_us": round(a["total_ms"] * 1e3 / a["calls"], 2)}
g = [v for k, v in out.items() if k.startswith("gemm_bf16_mfma")]
summary = {"families": out}
if g:
    n = sum(v["calls"] for v in g)
    summary["gemm_family_avg_launch_ms"] = round(sum(v["total_ms"] for v in g) / n, 5)
if len(sys.argv) > 3:
    steps = int(sys.argv[3])
    per_step = {"gemm_bf16_mfma": 99, "gemm_bf16_mfma_tn": 50}
    seen = collections.OrderedDict((k, []) for k in per_step)
    for r in csv.DictReader(open(sys.argv[2])):
        f = family(r["Kernel_Name"])
        if f in seen:
            seen[f].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    last = collections.OrderedDict()
    tot_ns, tot_n = 0, 0
    for f, d in seen.items():
        d.sort()
        d = d[-steps * per_step[f]:]
        ns = sum(e - b for b, e in d)
        last[f] = {"calls": len(d), "avg_us": round(ns / max(1, len(d)) / 1e3, 2)}
        tot_ns += ns
        tot_n += len(d)
    last["gemm_family_avg_launch_ms"] = round(tot_ns / max(1, tot_n) / 1e6, 5)
    summary["last_steps"] = {"steps": steps, **last}
print(json.dumps(summary, indent=1))
