#!/usr/bin/env python3
"""Randomised T2T-ViT (t2t.py:49-122): random image sizes, tokenizer layers (kernel, stride) -- i.e. tf.image.extract_patches 'SAME' geometries and
tokenizer transformers of odd widths (3 k^2, 3 k^2 k'^2, ...) -- backbone widths, pooling, batch (changing between calls on one object); logits,
every gradient and d(image) through the C ABI against oracle/ref_t2t.py.

    python tools/fuzz_t2t.py [n=20] [seed=0] [compute=fp32|bf16]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vit-tensorflow_amd")]


def run(n, seed, compute):
    from oracle import ref_t2t
    from vit_tensorflow.t2t import T2TViT
    rng = np.random.default_rng(seed)
    lowp = compute == "bf16"
    ltol, gtol = (1e-3, 1e-3) if not lowp else (3e-2, 9e-2)
    fails = []
    t0 = time.time()
    for i in range(n):
        layers = []
        pairs = [(3, 2), (7, 4), (5, 3), (4, 2), (5, 4), (6, 4), (3, 3), (4, 3), (2, 1), (3, 1)]   # mostly (kernel, stride) pairs the reference's own size formula accepts
        for _ in range(int(rng.integers(1, 4))):
            layers.append(pairs[int(rng.integers(0, len(pairs)))])
        size = int(rng.choice([16, 18, 20, 24, 27, 32, 36, 40, 48]))
        dim = 64 * int(rng.integers(1, 3)) if lowp else int(rng.choice([24, 32, 48]))
        kw = dict(image_size=size, num_classes=int(rng.choice([5, 17])), dim=dim, depth=int(rng.integers(1, 3)), heads=2, mlp_dim=2 * dim,
                  dim_head=64 if lowp else 16, pool="cls" if rng.random() < 0.5 else "mean", t2t_layers=tuple(layers))
        tag = f"#{i} {kw}"
        try:
            cfg = ref_t2t.make_config(**kw)
            same, sz = True, size
            for k, st in layers:   # t2t.py sizes pos_embedding with conv_output_size(size, k, stride, stride // 2) but the tokenizer ('SAME') yields ceil(size / stride)
                same = same and -(-sz // st) == ref_t2t.conv_output_size(sz, k, st, st // 2)
                sz = -(-sz // st)
            if not same:
                print(f"skip {tag}: the reference itself cannot run this (its pos_embedding has fewer rows than its tokenizer produces tokens)")
                continue
            if cfg["num_pos"] - 1 < 1 or cfg["num_pos"] > 300 or max(cfg["layer_dims"]) > 1400:
                print(f"skip {tag}: {cfg['num_pos'] - 1} positions, widths {cfg['layer_dims']}")
                continue
            P = ref_t2t.init_params(cfg, seed=50 + i)
            m = T2TViT(**kw, compute=compute, max_batch=3, seed=0)
            m.load_state_dict({k: v.astype(np.float32) for k, v in P.items()})
            worst = ("", 0.0)
            ok = True
            for b in (int(rng.integers(1, 4)), int(rng.integers(1, 4))):
                img = rng.standard_normal((b, size, size, 3)).astype(np.float32)
                dl = (rng.standard_normal((b, kw["num_classes"])) / b).astype(np.float32)
                logits = m(img, training=False)
                grads, dimg = m.backward(dl, want_dimg=True)
                rl, rg, rdimg = ref_t2t.forward_backward(cfg, P, img, dl, want_dimg=True)
                le = float(np.abs(logits - rl).max() / max(1.0, np.abs(rl).max()))
                errs = {k: float(np.abs(grads[k] - rg[k]).max() / (np.abs(rg[k]).max() + 1e-30)) for k in rg if np.asarray(rg[k]).size > (4 if lowp else 1)}
                errs["d(img)"] = float(np.abs(dimg - rdimg).max() / (np.abs(rdimg).max() + 1e-30))
                w = max(errs.items(), key=lambda kv: kv[1])
                worst = max(worst, w, key=lambda t: t[1])
                if not (np.isfinite(le) and le <= ltol and all(np.isfinite(v) and v <= gtol for v in errs.values())):
                    ok = False
                    print(f"FAIL {tag} b={b}: logits {le:.2e}, worst gradient {w[1]:.2e} ({w[0]})", flush=True)
            print(f"{'ok  ' if ok else 'FAIL'} {tag}: {cfg['num_pos'] - 1} positions, tokenizer widths {cfg['layer_dims']}, worst gradient {worst[1]:.2e} ({worst[0]})", flush=True)
            if not ok:
                fails.append(tag)
            del m
        except Exception as ex:
            print(f"FAIL {tag}: {type(ex).__name__}: {ex}", flush=True)
            fails.append(tag)
    print(f"{len(fails)} failures in {n} T2T-ViT configurations ({compute}; logits {ltol:g}, gradients {gtol:g}); {time.time() - t0:.0f} s")
    return fails


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 0, sys.argv[3] if len(sys.argv) > 3 else "fp32") else 0)
