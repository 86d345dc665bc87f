"""Build libvitx.so (the C-ABI HIP library), in-tree, for gfx950 (the oracle is numpy / torch: nothing of it is compiled).

    python vit-tensorflow_amd/build.py [--force]

hipcc cross-compiles without a GPU.  Objects go to vit-tensorflow_amd/build/, the library to
vit-tensorflow_amd/lib/libvitx.so (git-ignored; travels to the GPU box with the gpurun snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvitx.so")
SOURCES = ["env.hip", "elementwise.hip", "gemm_generic.hip", "gemm_f32_mfma.hip", "gemm_bf16x3.hip", "gemm_bf16.hip", "gemm_bf16_pipe.hip", "gemm_bf16_tn.hip", "attn_bf16.hip", "attn_x3.hip", "attn_generic.hip", "attn_bgemm_mfma.hip", "attn_headchain.hip", "attn_deepvit_fused.hip", "attn_cait_fused.hip", "mim_ops.hip", "mim.hip", "distill.hip", "comm.hip", "engine.hip", "capi.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# per-source additions.  attn_bf16.hip / attn_x3.hip: the running row maxima come straight out of MFMA accumulators; with NaNs honoured every fmaxf first
# canonicalises its operands (v_max_f32 x, x, x: 8 extra VALU instructions per key tile in a VALU-bound loop).  No NaN is produced or
# consumed on that path (masked keys are -inf selects, never arithmetic on NaN).
# gemm_bf16_tn.hip: the machine scheduler's max-ILP strategy orders the transpose reads / MFMAs of the weight-gradient K loop 1.6 % faster
# (8.06 -> 7.93 ms per step as a class, `profiles/r4/ab_gemm_sched_strategy_max_ilp_r4st.log`; the NT kernel does not move and one of its
# instantiations then fails tools/isa_check.py, so it keeps the default).
EXTRA_FLAGS = {"attn_bf16.hip": ["-fno-honor-nans"], "attn_x3.hip": ["-fno-honor-nans", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],   # (attn_x3: -0.3 ms per BF16X3 step)
               "gemm_bf16_tn.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
               # gemm_bf16x3.hip (VALU-bound split arithmetic around three MFMAs per product): the max-memory-clause strategy, BF16X3 step at batch 256
               # 108.9 -> 105.5 ms on the same box (max-ilp 107.4, iterative-minreg 106.9; `profiles/r4/ab_bf16x3_gemm_sched_strategies_r4st.log`)
               "gemm_bf16x3.hip": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]}


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "vitx.h"))
    hdrs.append(os.path.abspath(__file__))   # FLAGS / EXTRA_FLAGS live here: an edited flag set rebuilds every object
    return max(os.path.getmtime(h) for h in hdrs)


def flags_id() -> str:
    """Digest of the compile flags (part of kernel_source_id(): tuned GEMM profiles / PMC traffic files are refused when it changes)."""
    import hashlib
    return hashlib.sha256(repr((FLAGS, sorted(EXTRA_FLAGS.items()))).encode()).hexdigest()[:12]


# Objects whose K loops issue LDS-DMA from inline asm and keep M0 alive between the pieces of a group (common.h, vitx_dma16 / vitx_dma16_cont):
# hipcc does not model M0 across asm statements (an "m0" clobber only draws `inline asm clobber list contains reserved registers`), so the
# guarantee is checked on the ISA of EVERY build instead of trusted: no M0 write but the DMA statements' own, no compiler vmcnt wait and no
# scratch access inside a K loop (tools/isa_check.py).  A compiler that breaks either fails the build here, not a run on the GPU.
ISA_CHECKED = ["gemm_bf16_pipe.hip", "gemm_bf16_tn.hip"]


def _isa_gate(objs) -> None:
    import importlib.util
    tool = os.path.join(HERE, "..", "tools", "isa_check.py")
    if not os.path.exists(tool):
        return
    spec = importlib.util.spec_from_file_location("vitx_isa_check", tool)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for src, obj in objs:
        stamp = obj + ".isa_ok"
        if os.path.exists(stamp) and os.path.getmtime(stamp) >= os.path.getmtime(obj):
            continue
        bad = []
        for name, body in mod.kernels(mod.disassemble(obj)):
            if "nt_pipe_kernel" not in name and "gemm_bf16_tn_kernel" not in name:
                continue
            r = mod.check(name, body)
            if r is not None and not r[0]:
                bad.append(r[1])
        if bad:
            raise RuntimeError(f"ISA check failed for {src} (tools/isa_check.py):\n" + "\n".join(bad))
        open(stamp, "w").write("ok\n")


ASAN_FLAGS = ["-fsanitize=address", "-shared-libsan", "-fno-gpu-sanitize", "-g"]   # host code only (the device side needs xnack+ targets)
ASAN_RT = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"


def _compile(src: str, force: bool, asan: bool = False) -> str:
    if asan:
        obj = os.path.join(HERE, "build_asan", src.replace(".hip", ".o"))
        s = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(s), _deps_mtime()):
            return obj
        r = subprocess.run([HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), *ASAN_FLAGS, "-c", s, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc (asan) failed for {src}:\n{r.stderr}")
        return obj
    obj = os.path.join(BUILD, src.replace(".hip", ".o"))
    s = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(s), _deps_mtime()):
        return obj
    cmd = [HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
    return obj


def build(force: bool = False) -> str:
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    _isa_gate([(s, o) for s, o in zip(SOURCES, objs) if s in ISA_CHECKED])
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


def build_asan(force: bool = False) -> str:
    """libvitx_asan.so: the same sources with the HOST side under AddressSanitizer (C-ABI argument handling, parameter tables, engine
    bookkeeping).  Load it with LD_PRELOAD=ASAN_RT and VITX_LIB=<this file> (tests/test_asan.py)."""
    os.makedirs(os.path.join(HERE, "build_asan"), exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    lib = os.path.join(LIBDIR, "libvitx_asan.so")
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, True), SOURCES))
    if force or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *ASAN_FLAGS, "-o", lib, *objs, "-ldl"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link (asan) failed:\n{r.stderr}")
    return lib


def build_diag(force: bool = False) -> str:
    """libvitx_diag.so: the release objects with csrc/env.hip recompiled under -DVITX_DIAG, i.e. the library that HONOURS the diagnostic
    environment switches (timing experiments that may corrupt results; csrc/env.h).  The release library ignores them.  Load with VITX_LIB=<this file>."""
    build(force)
    lib = os.path.join(LIBDIR, "libvitx_diag.so")
    os.makedirs(os.path.join(HERE, "build_diag"), exist_ok=True)
    obj = os.path.join(HERE, "build_diag", "env.o")
    src = os.path.join(CSRC, "env.hip")
    if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), _deps_mtime()):
        r = subprocess.run([HIPCC, *FLAGS, "-DVITX_DIAG=1", "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc (diag) failed:\n{r.stderr}")
    objs = [os.path.join(BUILD, s_.replace(".hip", ".o")) for s_ in SOURCES if s_ != "env.hip"] + [obj]
    if force or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link (diag) failed:\n{r.stderr}")
    return lib


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_asan(force) if "--asan" in sys.argv else build_diag(force) if "--diag" in sys.argv else build(force))
