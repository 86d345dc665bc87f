#!/usr/bin/env python3
"""MFMA-busy / wait / LDS-conflict shares per kernel family and per NT-GEMM launch shape from two `rocprofv3 --pmc` passes over
`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile` (the default ViT-B/16 step):

    python tools/pmc_sq_summary.py <pass_insts counter_collection.csv> <pass_waits counter_collection.csv> > profiles/rN/pmc_sq_summary.txt

pass_insts: SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE
pass_waits: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES

Only the LAST step's dispatches are used (the first step also runs the per-shape variant measurement).  Derived numbers:
  MFMA busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)     -- share of SIMD-cycles with the matrix pipe busy
  wait any   = SQ_WAIT_ANY / SQ_WAVE_CYCLES, wait inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES -- share of wave-cycles spent waiting
  LDS confl. = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
The NT-GEMM launches of a step are labelled by (tile template, epilogue mode, MFMA instruction count): a 32x32x16 instruction is 32768
FLOP, so the count is 2 M N K / 32768 (padded to whole tiles) and launches of equal size and epilogue share a row."""
import collections
import csv
import re
import sys

FAMILIES = ["gemm_bf16_nt", "gemm_bf16_tn", "attn_fwd", "attn_bwd", "layernorm_fwd", "layernorm_bwd", "reduce_partials"]
LAST = {"gemm_bf16_nt": 99, "gemm_bf16_tn": 50, "attn_fwd": 12, "attn_bwd": 12, "layernorm_fwd": 25, "layernorm_bwd": 25, "reduce_partials": 61}
MODES = {"0": "bf16 store", "1": "fp32 store", "2": "bias + GELU (act, gelu')", "3": "bias + fp32 residual", "4": "patch embed", "5": "x gelu' + column sums", "6": "partials"}   # epilogue.h: EpiMode


def load(path):
    """family -> ordered list of dispatches (dict counter -> value, plus name) of the last step"""
    disp = collections.OrderedDict()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"]})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = {}
    for fam in FAMILIES:
        out[fam] = [d for d in disp.values() if fam in d["name"]][-LAST[fam]:]
    return out


def tot(ds, k):
    return sum(d.get(k, 0.0) for d in ds)


def line(label, n, a, b):
    gui = tot(a, "GRBM_GUI_ACTIVE") / 8.0
    wc = tot(b, "SQ_WAVE_CYCLES")
    idx = tot(a, "SQ_LDS_IDX_ACTIVE")
    busy = tot(a, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * gui) if gui else 0.0
    cyc = gui / n if n else 0.0
    s = f"{label:<58s} {n:3d} launches  {cyc / 1e3:8.1f} kcycles each   MFMA busy {100 * busy:5.1f} %"
    if wc:
        s += f"   wait any {100 * tot(b, 'SQ_WAIT_ANY') / wc:5.1f} %  wait inst {100 * tot(b, 'SQ_WAIT_INST_ANY') / wc:5.1f} %"
    if idx:
        s += f"   LDS conflict {100 * tot(a, 'SQ_LDS_BANK_CONFLICT') / idx:5.1f} %"
    return s


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    print("# per kernel family, last step of `bench.py --steps 1 --warmup 1` (ViT-B/16, batch 256)")
    for fam in FAMILIES:
        print(line(fam, len(a[fam]), a[fam], b[fam] if len(b[fam]) == len(a[fam]) else []))
    print("\n# NT GEMM launches of that step by (tile <BM,BN,WM,WN>, epilogue, size = SQ_INSTS_MFMA x 32768 FLOP, tile padding included), slowest class first")
    groups = collections.OrderedDict()
    same = len(a["gemm_bf16_nt"]) == len(b["gemm_bf16_nt"])
    for i, d in enumerate(a["gemm_bf16_nt"]):
        m = re.search(r"gemm_bf16_nt\w*<([^>]*)>", d["name"])
        targs = m.group(1).replace(" ", "") if m else "?"
        parts = targs.split(",")
        mode = parts[4] if len(parts) > 4 else "?"     # <BM, BN, WM, WN, MODE, ...>
        size = round(d.get("SQ_INSTS_MFMA", 0.0) / 1e6, 1)
        key = (",".join(parts[:4]), mode, size)
        g = groups.setdefault(key, ([], []))
        g[0].append(d)
        if same:
            g[1].append(b["gemm_bf16_nt"][i])
    for (tile, mode, size), (da, db) in sorted(groups.items(), key=lambda kv: -tot(kv[1][0], "GRBM_GUI_ACTIVE")):
        gflop = size * 1e6 * 32768 / 1e9
        print(line(f"<{tile}> {MODES.get(mode, mode)}, {gflop:6.1f} GFLOP", len(da), da, db))


if __name__ == "__main__":
    main()
