#!/usr/bin/env python3
"""Same-box A/B of two builds of libvitx.so.

The boxes of the GPU pool differ by up to ~9 % on the epilogue-heavy GEMM launches, so a kernel change worth a
few per cent can only be judged inside ONE gpurun call.  Keep a copy of the baseline library, rebuild, then:

    cp vit-tensorflow_amd/lib/libvitx.so vit-tensorflow_amd/lib/libvitx_base.so      # before the change
    python vit-tensorflow_amd/build.py                                                 # after the change
    gpurun -- python tools/ab_bench.py vit-tensorflow_amd/lib/libvitx_base.so vit-tensorflow_amd/lib/libvitx.so [rounds] [-- bench args]

bench.py runs alternately with VITX_LIB pointing at A and B (A B A B ...); per kernel class the script prints the mean ms per step
of each side and the difference.  (Files matching *.so are git-ignored but travel with the gpurun snapshot.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    if len(argv) < 2:
        raise SystemExit(__doc__)
    libs = {"A": os.path.abspath(argv[0]), "B": os.path.abspath(argv[1])}
    rounds = int(argv[2]) if len(argv) > 2 else 3
    res = {"A": [], "B": []}
    for r in range(rounds):
        for side in ("A", "B"):
            env = dict(os.environ, VITX_LIB=libs[side])
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "10", "--warmup", "3", *extra],
                               env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                raise SystemExit(f"bench.py failed for side {side}:\n{p.stderr[-2000:]}")
            res[side].append(json.loads(line[-1]))
            print(f"round {r} {side}: {res[side][-1]['ms_per_step']:.3f} ms", flush=True)
    mean = lambda xs: sum(xs) / len(xs)
    out = {"A": libs["A"], "B": libs["B"], "ms_per_step": {s: [d["ms_per_step"] for d in res[s]] for s in res}, "classes": {}}
    names = sorted({k for s in res for d in res[s] for k in d.get("kernel_classes", {})})
    for k in names:
        a = mean([d["kernel_classes"].get(k, {}).get("ms_per_step", 0.0) for d in res["A"]])
        b = mean([d["kernel_classes"].get(k, {}).get("ms_per_step", 0.0) for d in res["B"]])
        if max(a, b) >= 0.05:
            out["classes"][k] = {"A_ms": round(a, 4), "B_ms": round(b, 4), "B_minus_A_ms": round(b - a, 4)}
    out["step"] = {"A_ms": round(mean(out["ms_per_step"]["A"]), 3), "B_ms": round(mean(out["ms_per_step"]["B"]), 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
