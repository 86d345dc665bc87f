#!/usr/bin/env python3
"""Per-queue view of one step of a rocprofv3 kernel trace (CSV with the rows of ONE step):
   python tools/trace_streams.py <kernel_trace_last_step.csv>
Prints, per hardware queue, the launches and busy time, the idle gaps of the main queue (largest first, with the kernels around them) and
the time during which no kernel of any queue runs."""
import collections
import csv
import re
import sys


def short(n):
    return re.sub(r"<.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "").replace("_ZN12_GLOBAL__N_1", ""))[:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    t0 = rows[0]["s"]
    step = rows[:-1] if "unfold" in rows[-1]["Kernel_Name"] else rows
    span = (rows[-1]["s"] - t0) / 1e6
    print(f"step span {span:.3f} ms, {len(step)} launches")
    byq = collections.defaultdict(list)
    for r in step:
        byq[r["Queue_Id"]].append(r)
    mainq = max(byq, key=lambda q: len(byq[q]))
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(r["e"] - r["s"] for r in rs) / 1e6
        names = collections.Counter(short(r["Kernel_Name"]) for r in rs).most_common(4)
        print(f"queue {q}{' (main)' if q == mainq else ''}: {len(rs)} launches, {busy:.3f} ms of kernels, first start {(rs[0]['s'] - t0) / 1e6:.2f} ms, last end {(max(r['e'] for r in rs) - t0) / 1e6:.2f} ms; {names}")
    m = byq[mainq]
    gaps = [(m[i + 1]["s"] - max(x["e"] for x in m[: i + 1][-3:]), i) for i in range(len(m) - 1)]
    gaps = [(g, i) for g, i in gaps if g > 0]
    print(f"main queue: {len(gaps)} idle gaps, {sum(g for g, _ in gaps) / 1e6:.3f} ms in total")
    agg = collections.defaultdict(lambda: [0, 0])
    for g, i in gaps:
        k = (short(m[i]["Kernel_Name"]), short(m[i + 1]["Kernel_Name"]))
        agg[k][0] += 1
        agg[k][1] += g
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"   {v[1] / 1e3:8.1f} us in {v[0]:3d} gaps   after {k[0]:44s} before {k[1]}")
    ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
    depth, prev, idle = 0, ev[0][0], 0
    for t, d in ev:
        if depth == 0:
            idle += t - prev
        depth += d
        prev = t
    print(f"no kernel of any queue running: {idle / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
