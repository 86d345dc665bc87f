#!/usr/bin/env python3
"""Register / LDS / spill metadata per kernel of a HIP object (the code object's msgpack notes):
    python tools/kernel_meta.py <object.o> [name-substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def notes(obj: str) -> str:
    with tempfile.TemporaryDirectory() as td:
        co, fb = f"{td}/dev.co", f"{td}/fat.bin"
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj, f"{td}/copy.o"], capture_output=True, text=True, check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fb}", f"--output={co}", "--unbundle"],
                       capture_output=True, text=True, check=True)
        return subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout


def main():
    obj, subs = sys.argv[1], sys.argv[2:]
    cur = {}
    for line in notes(obj).splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if line.lstrip().startswith("- .") and k not in ("name",) and "name" in cur and k == "agpr_count":
            pass
        cur[k] = v
        if k == "wavefront_size":   # last key of a kernel record
            name = cur.get("name", "?")
            if not subs or all(s in name for s in subs):
                print(f"{name[:110]}: vgpr {cur.get('vgpr_count')} agpr {cur.get('agpr_count')} sgpr {cur.get('sgpr_count')} lds {cur.get('group_segment_fixed_size')} "
                      f"scratch {cur.get('private_segment_fixed_size')} spill v{cur.get('vgpr_spill_count')} s{cur.get('sgpr_spill_count')} max_wg {cur.get('max_flat_workgroup_size')}")
            cur = {}


if __name__ == "__main__":
    main()
