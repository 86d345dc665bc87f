#!/usr/bin/env python3
"""Phase timeline of co-resident GEMM workgroups from the cycle stamps of a diagnostic build (tools/build_variant.sh stamps
"-DVITX_GEMM_STAMPS_BUILD=1" gemm_bf16_pipe.hip; VITX_LIB=.../libvitx_stamps.so VITX_GEMM_STAMPS=2 python tools/gemm_bench.py M N K variant epi 1):

    python tools/stamps_timeline.py <stderr log> [CUs to print]

Groups the workgroups by the CU they ran on (XCC_ID, HW_ID se/sh/cu bits) and prints, per CU, each workgroup's K-loop and epilogue intervals in
thousands of shader-clock cycles from the launch's first stamp (the counter runs at the shader clock, ~1.8-2.1 GHz under these loads), plus the fraction of the CU's busy span during which a K loop of one
workgroup ran beside an epilogue of another (the overlap two workgroups per CU are there for)."""
import re
import sys
from collections import defaultdict

TICK_US = 0.001   # printed unit = 1000 cycles of the shader clock counter (__builtin_readcyclecounter)


def main():
    path = sys.argv[1]
    ncu = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cus = defaultdict(list)
    t0 = None
    for line in open(path):
        m = re.match(r"\[stamps-raw\] wg (\d+) xcc (\d+) hwid (0x[0-9a-f]+) \| (.*)", line.strip())
        if not m:
            continue
        wg, xcc, hw = int(m.group(1)), int(m.group(2)), int(m.group(3), 16)
        tiles = [[int(x) for x in t.split()] for t in m.group(4).split(" | ")]
        key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
        cus[key].append((wg, tiles))
        lo = min(t[0] for t in tiles)
        t0 = lo if t0 is None else min(t0, lo)
    if not cus:
        print("no [stamps-raw] rows"); return
    occ = defaultdict(int)
    for v in cus.values():
        occ[len(v)] += 1
    print("workgroups per CU (CUs):", dict(sorted(occ.items())))
    tot_overlap = tot_span = 0.0
    shown = 0
    for key in sorted(cus):
        wgs = cus[key]
        ev = []   # (time, kind, +1/-1)
        for wg, tiles in wgs:
            for t in tiles:
                ev += [(t[0], "K", 1), (t[1], "K", -1), (t[1], "E", 1), (t[2], "E", -1)]
        ev.sort()
        k = e = 0
        last = ev[0][0]
        both = 0
        for tm, kind, d in ev:
            if k > 0 and e > 0:
                both += tm - last
            last = tm
            if kind == "K": k += d
            else: e += d
        span = ev[-1][0] - ev[0][0]
        tot_overlap += both; tot_span += span
        if shown < ncu and len(wgs) >= 2:
            shown += 1
            t0 = ev[0][0]   # the cycle counters of different XCDs are not aligned: times are from this CU's first stamp
            print(f"CU xcc {key[0]} se {key[1]} sh {key[2]} cu {key[3]}: {len(wgs)} workgroups, K-loop || epilogue overlap {both * TICK_US:.1f} of {span * TICK_US:.1f} kcycles")
            for wg, tiles in wgs:
                print(f"   wg {wg:3d}: " + "  ".join(f"K {(t[0] - t0) * TICK_US:6.1f}-{(t[1] - t0) * TICK_US:6.1f} E -{(t[2] - t0) * TICK_US:6.1f}" for t in tiles))
    print(f"all CUs: a K loop ran beside another workgroup's epilogue during {100.0 * tot_overlap / max(tot_span, 1):.1f} % of the CUs' busy spans")


if __name__ == "__main__":
    main()
