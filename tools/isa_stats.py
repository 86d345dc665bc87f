#!/usr/bin/env python3
"""Static instruction mix per kernel of a HIP object:  python tools/isa_stats.py <object.o> [name-substring ...]
(VALU / MFMA / LDS / SALU / s_nop / waits; a quick check of what a source change did to a VALU-bound kernel)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_check  # noqa: E402


def main():
    obj, subs = sys.argv[1], sys.argv[2:]
    for name, body in isa_check.kernels(isa_check.disassemble(obj)):
        if subs and not all(s in name for s in subs):
            continue
        c = collections.Counter()
        for l in body:
            op = l.split()[0]
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith("v_"): c["valu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith("s_nop"): c["s_nop"] += 1
            elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
            elif op.startswith("s_cbranch") or op.startswith("s_branch"): c["branch"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c["vmem"] += 1
            else: c["other"] += 1
        print(f"{name[:100]}: " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()
