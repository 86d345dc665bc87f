#!/usr/bin/env python3
"""Which kernels of a step actually run at the same time?  From a rocprofv3 --kernel-trace CSV (last `steps` steps of the run):
    python tools/overlap_trace.py <kernel_trace.csv> [fraction of the trace to keep from the end, default 0.3]
For every kernel family: launches, summed duration, and the part of that duration during which a kernel of ANOTHER queue was running (split by the
other kernel's family).  A LayerNorm VJP that co-resides with a weight-gradient GEMM shows up as `layernorm_bwd ... under gemm_tn NN %`."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import family  # noqa: E402


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    keep = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    for r in rows:
        r["s"], r["e"], r["f"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), family(r["Kernel_Name"])
    rows.sort(key=lambda r: r["s"])
    t_end = max(r["e"] for r in rows)
    t_beg = rows[0]["s"]
    cut = t_end - (t_end - t_beg) * keep
    rows = [r for r in rows if r["s"] >= cut]
    span = (max(r["e"] for r in rows) - rows[0]["s"]) / 1e6
    print(f"{len(rows)} launches over {span:.2f} ms; queues: {sorted(set(r['Queue_Id'] for r in rows))}")
    dur = collections.Counter()
    cnt = collections.Counter()
    ov = collections.defaultdict(collections.Counter)
    for i, r in enumerate(rows):
        dur[r["f"]] += r["e"] - r["s"]
        cnt[r["f"]] += 1
        for o in rows[max(0, i - 40): i + 40]:
            if o is r or o["Queue_Id"] == r["Queue_Id"]:
                continue
            lo, hi = max(r["s"], o["s"]), min(r["e"], o["e"])
            if hi > lo:
                ov[r["f"]][o["f"]] += hi - lo
    for f, d in dur.most_common(14):
        parts = ", ".join(f"{g} {100.0 * v / d:.0f} %" for g, v in ov[f].most_common(3))
        print(f"{f:28s} {cnt[f]:4d} launches {d / 1e6:8.3f} ms (avg {d / cnt[f] / 1e3:7.1f} us)   concurrent with another queue's: {parts or '-'}")


if __name__ == "__main__":
    main()
