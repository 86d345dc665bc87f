"""HBM roofline of the T2T tokenizer kernel (vitx_extract_patches_dev, t2t.py:42): algorithmic bytes = read x once + write out once,
timed with torch events on the default stream (the kernel is launched on stream 0).  usage: python tools/bench_extract_patches.py [b]"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vit-tensorflow_amd")]
from vit_tensorflow import _native as N  # noqa: E402
from vit_tensorflow import t2t  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = N.lib()
dev = torch.device("cuda", 0)
out = []
for H, Cc, k, s in ((224, 3, 7, 4), (56, 147, 3, 2), (28, 1323, 3, 2)):      # T2TViT's three tokenizer steps at 224 px (t2t.py:52,62-70)
    bb = b if Cc < 1000 else max(1, b // 8)
    oh, ow, f = t2t.extract_patches_shape(H, H, Cc, k, s)
    x = torch.randn(bb, H, H, Cc, device=dev)
    y = torch.empty(bb, oh, ow, f, device=dev)
    dx = torch.empty_like(x)
    res = {"geometry": f"b={bb} {H}x{H}x{Cc} k={k} s={s} -> {oh}x{ow}x{f}"}
    for name, fn, args in (("fwd", lib.vitx_extract_patches_dev, (x, y)), ("bwd", lib.vitx_extract_patches_backward_dev, (y, dx))):
        call = lambda: N.check(fn(C.c_void_p(args[0].data_ptr()), bb, H, H, Cc, k, s, C.c_void_p(args[1].data_ptr()), None))
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        byts = (x.numel() + y.numel()) * 4
        res[name] = {"ms": round(ms, 4), "algorithmic_GB": round(byts / 1e9, 4), "GBps": round(byts / ms / 1e6, 1), "frac_of_8TBps": round(byts / ms / 1e6 / 8000, 3)}
    out.append(res)
print(json.dumps(out, indent=1))
