#!/usr/bin/env python3
"""Same-box A/B of two (or more) environment settings of ONE build of libvitx.so.

    gpurun -- python tools/ab_env.py "VITX_SIDE_STREAM=0" "VITX_SIDE_STREAM=1" [more settings ...] [--rounds 3] [-- bench args]

bench.py runs alternately under each setting (A B C A B C ...); prints ms per step of every run and the mean per setting.  A setting is a
space-separated list of NAME=VALUE pairs ("" = the plain environment).  Boxes of the pool differ by several per cent, so only differences
inside one call mean anything."""
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
    rounds = 3
    if "--rounds" in argv:
        i = argv.index("--rounds")
        rounds = int(argv[i + 1])
        del argv[i:i + 2]
    if len(argv) < 2:
        raise SystemExit(__doc__)
    res = {s: [] for s in argv}
    for r in range(rounds):
        for s in argv:
            env = dict(os.environ)
            for kv in s.split():
                k, v = kv.split("=", 1)
                env[k] = v
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-profile", "--steps", "20", "--warmup", "3", *extra],
                               env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                raise SystemExit(f"bench.py failed under '{s}':\n{p.stderr[-2000:]}")
            d = json.loads(line[-1])
            res[s].append(d["ms_per_step"])
            print(f"round {r} [{s}]: {d['ms_per_step']:.3f} ms  {d['value']:.1f} {d['unit']}", flush=True)
    print(json.dumps({"bench_args": extra, "ms_per_step": res, "mean_ms": {s: round(sum(v) / len(v), 3) for s, v in res.items()}}, indent=1))


if __name__ == "__main__":
    main()
