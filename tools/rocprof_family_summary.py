#!/usr/bin/env python3
"""Fold a rocprofv3 `--kernel-trace --stats` kernel_stats.csv into kernel families (all template instantiations of the NT / TN
GEMM, LayerNorm, attention, ...): calls, total and average duration.  usage: rocprof_family_summary.py <kernel_stats.csv>
Note: the first step of a process also contains the GEMM variant measurements (a few launches of each candidate)."""
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
print(json.dumps(summary, indent=1))
