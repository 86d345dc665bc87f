#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel family:  python tools/pmc_sum.py <counter_collection.csv> [name-substring ...]"""
import csv
import sys
from collections import defaultdict

path, subs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(set)
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"]
        fam = next((s for s in subs if s in name), None) if subs else name[:50]
        if fam is None:
            continue
        acc[fam][row["Counter_Name"]] += float(row["Counter_Value"])
        n[fam].add(row["Dispatch_Id"])
for fam, c in acc.items():
    print(f"{fam} ({len(n[fam])} dispatches): " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
        print(f"    LDS bank-conflict share of LDS cycles: {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f}")
