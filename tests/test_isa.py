"""CPU tier: build-time ISA properties of the pipelined bf16 GEMM kernels (tools/isa_check.py; ADVICE r3 on vitx_dma16 / M0).

The K loops issue their LDS-DMA from inline asm the compiler cannot see and rely on (1) no compiler-inserted `s_waitcnt vmcnt` inside the K loop
(it would drain the DMA in flight: scratch reloads and epilogue loads left pending across the tile loop's back edge produce them -- round 3's
320-row residual variant shipped with four) and (2) nothing but the DMA statements writing M0.  Checked on the objects the library is linked from."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "vit-tensorflow_amd", "build")


def test_pipelined_gemm_kernels_have_one_vmcnt_wait_per_k_tile_and_no_foreign_m0_writes():
    objs = [os.path.join(BUILD, f) for f in ("gemm_bf16_pipe.o", "gemm_bf16_tn.o")]
    if not all(os.path.exists(o) for o in objs):   # library came pre-built without its objects: rebuild them (hipcc cross-compiles, ~1 min)
        sys.path.insert(0, os.path.join(ROOT, "vit-tensorflow_amd"))
        import build as _b
        _b.build(force=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_check.py"), *objs], capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) >= 16, r.stdout + r.stderr          # 7 epilogues x 2 tile heights of the NT kernel + 2 tile sizes of the weight-gradient kernel
    assert r.returncode == 0 and all(l.startswith("ok") for l in lines), "\n".join(l for l in lines if not l.startswith("ok")) + r.stderr
