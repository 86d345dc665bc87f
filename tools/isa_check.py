#!/usr/bin/env python3
"""Build-time ISA check of the pipelined bf16 GEMM kernels (tests/test_isa.py runs it on the objects of the built library).

The K loops of gemm_bf16_nt_pipe_kernel / gemm_bf16_tn_kernel issue their LDS-DMA from inline asm the compiler's s_waitcnt pass cannot see,
and rely on two things no compiler pass guarantees:
  1. NO compiler-inserted `s_waitcnt vmcnt` inside the K loop: at run time such a wait also drains the DMA pieces in flight (hardware vmcnt
     counts them) -- the loop then runs at the speed of un-prefetched loads.  Scratch reloads, epilogue loads left "pending" across the tile
     loop's back edge and LDS-DMA builtins all produce them.  Exactly ONE vmcnt wait per K-tile iteration is the schedule's own (hand-over).
  2. M0 is written ONLY by the asm DMA statements (`s_mov_b32 m0, sN` directly followed by `s_nop` + `buffer_load ... lds`): continuation
     pieces reuse the M0 value of the group's first piece across MFMAs, barriers and epilogue code.

    python tools/isa_check.py <object.o> [...]        # prints one line per kernel, exit code 1 on a violation
"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj: str) -> str:
    with tempfile.TemporaryDirectory() as td:
        co, fb = f"{td}/dev.co", f"{td}/fat.bin"
        # the fat binary of a HIP object; the explicit output file keeps llvm-objcopy from rewriting `obj` in place (which would bump its
        # mtime and make build.py skip the next recompile of an edited source)
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj, f"{td}/copy.o"], capture_output=True, text=True, check=True)
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fb}", f"--output={co}", "--unbundle"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout


def kernels(asm: str):
    cur, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
        elif cur and line.strip():
            body.append(line.strip())
    if cur:
        yield cur, body


def check(name: str, body):
    ins = [re.sub(r"\s*//.*", "", l) for l in body]
    mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
    if not mf:
        return None
    lo, hi = mf[0], mf[-1]
    # the K loop = first to last MFMA (the epilogue holds none)
    waits = [l for l in ins[lo:hi + 1] if l.startswith("s_waitcnt") and "vmcnt" in l]
    scratch = [l for l in ins[lo:hi + 1] if l.startswith("scratch_")]
    bad_m0 = []
    for i, l in enumerate(ins):
        ops = l.split(None, 1)
        dst = ops[1].split(",")[0].strip() if len(ops) > 1 else ""
        if dst == "m0" or l.startswith(("s_movrel", "s_set_gpr_idx")):
            nxt = ins[i + 1:i + 3]
            ok = l.startswith("s_mov_b32 m0, s") and len(nxt) == 2 and nxt[0].startswith("s_nop") and nxt[1].startswith("buffer_load_dwordx4") and nxt[1].rstrip().endswith("lds")
            if not ok:
                bad_m0.append(l)
    ok = len(waits) == 1 and not scratch and not bad_m0
    return ok, f"{'ok  ' if ok else 'FAIL'} vmcnt waits in K loop {len(waits)} (want 1), scratch ops in K loop {len(scratch)}, foreign M0 writes {len(bad_m0)}  {name[:110]}"


def main():
    rc = 0
    for obj in sys.argv[1:]:
        for name, body in kernels(disassemble(obj)):
            if "nt_pipe_kernel" not in name and "gemm_bf16_tn_kernel" not in name:
                continue
            r = check(name, body)
            if r is None:
                continue
            print(r[1])
            rc |= 0 if r[0] else 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
