#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py` to HBM bytes per launch of each kernel family.

    python tools/pmc_traffic.py <dir with pmc_FETCH_SIZE*/ and pmc_WRITE_SIZE*/> <launches of one step> > profiles/rN/pmc_traffic.json

Only the LAST step's dispatches are kept (the first step of a process also carries the GEMM variant measurements).
Corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB-like units of 1024 B as reported by rocprofv3 on gfx950;
FETCH_SIZE tallies 128-B requests at 64 B, so fetched bytes = 2 x FETCH_SIZE for the wide coalesced reads these kernels issue;
WRITE_SIZE is taken as reported (uncalibrated -- stated in the output)."""
import csv
import glob
import json
import os
import sys
from collections import OrderedDict, defaultdict


def family(name: str) -> str:
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    if "gemm_bf16_tn" in n:
        return "gemm_bf16_mfma_tn"
    if "gemm_bf16_nt" in n:
        return "gemm_bf16_mfma"
    n = n.split("<")[0].strip()
    if n.startswith("_Z"):   # mangled: keep the function identifier only
        import re
        m = re.search(r"\d+([a-z][a-z0-9_]*_kernel)", n)
        n = m.group(1) if m else n
    return n


def load(root: str, counter: str):
    files = glob.glob(os.path.join(root, f"pmc_{counter}*", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return None
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def main():
    # usage: pmc_traffic.py <dir> [family=launches_per_step,...]   (families not listed: last half of their dispatches, the
    # profiled command runs two steps)
    root = sys.argv[1]
    last = {}
    if len(sys.argv) > 2:
        for kv in sys.argv[2].split(","):
            k, v = kv.split("=")
            last[k] = int(v)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-tensorflow_amd"))
    from vit_tensorflow._model_math import kernel_source_id
    out = OrderedDict(kernel_source_id=kernel_source_id(), source="rocprofv3 --pmc (separate passes) over bench.py --steps 1 --warmup 1, LAST step only",
                      corrections="fetch bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE x 1024 as reported (uncalibrated)")
    fams = defaultdict(lambda: {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
    for counter, key, mult in (("FETCH_SIZE", "fetch_bytes", 2.0 * 1024.0), ("WRITE_SIZE", "write_bytes", 1024.0)):
        rows = load(root, counter)
        if rows is None:
            out[f"missing_{counter}"] = True
            continue
        byfam = defaultdict(list)
        for _, name, v in rows:
            byfam[family(name)].append(v)
        for fam, vals in byfam.items():
            keep = last.get(fam, max(1, len(vals) // 2))
            vals = vals[-keep:]
            f = fams[fam]
            f[key] += sum(vals) * mult
            if counter == "FETCH_SIZE":
                f["launches"] += len(vals)
    res = OrderedDict()
    for k, f in sorted(fams.items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"])):
        n = max(f["launches"], 1)
        res[k] = {"launches_per_step": f["launches"], "hbm_bytes_per_launch": round((f["fetch_bytes"] + f["write_bytes"]) / n),
                  "fetch_bytes_per_launch": round(f["fetch_bytes"] / n), "write_bytes_per_launch": round(f["write_bytes"] / n)}
    out["families"] = res
    g = [v for k, v in res.items() if k.startswith("gemm_bf16_mfma")]
    if g:
        n = sum(v["launches_per_step"] for v in g)
        out["gemm_family_hbm_bytes_per_launch"] = round(sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for v in g) / max(n, 1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
