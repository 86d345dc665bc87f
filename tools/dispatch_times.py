#!/usr/bin/env python3
"""Per-shape launch durations from a rocprofv3 --kernel-trace CSV: groups the dispatches of kernels whose name contains a given
substring by grid size and prints count / mean / min duration (us).

    python tools/dispatch_times.py <kernel_trace.csv> <name-substring> [skip_first_n]"""
import csv
import sys
from collections import defaultdict

path, sub = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
groups = defaultdict(list)
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        name = row.get("Kernel_Name", "")
        if sub not in name:
            continue
        key = (name[:60], row.get("Grid_Size_X") or row.get("Grid_Size"), row.get("Workgroup_Size_X") or row.get("Workgroup_Size"))
        groups[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for key, d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    d = d[skip:] if len(d) > skip else d
    print(f"{key[0]:60s} grid {key[1]:>8s} wg {key[2]:>4s}: n {len(d):4d}  mean {sum(d) / len(d):9.2f} us  min {min(d):9.2f} us  total {sum(d) / 1e3:8.3f} ms")
