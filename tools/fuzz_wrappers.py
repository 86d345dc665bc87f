#!/usr/bin/env python3
"""Randomised MAE / SimMIM wrappers (mae.py:14-92, simmim.py:62-130) around random ViT / DeepViT encoders: random image / patch sizes, masking
ratios (one masked patch .. all but one), decoder widths, batch; loss, predicted pixel values and every gradient through the C ABI against
oracle/ref_wrappers.py on identical weights, images and indices.

    python tools/fuzz_wrappers.py [n=30] [seed=0] [compute=fp32|bf16]

tests/test_gpu_fuzz.py runs a fixed-seed slice."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vit-tensorflow_amd")]


def _rel(a, ref, norm=False):
    ref = np.asarray(ref, np.float64)
    d = np.asarray(a, np.float64).reshape(ref.shape) - ref
    if norm:
        return float(np.linalg.norm(d) / max(1e-30, np.linalg.norm(ref)))
    return float(np.abs(d).max() / max(1e-6, np.abs(ref).max()))


def run(n, seed, compute):
    from oracle import ref_torch, ref_wrappers as RW, spec
    from vit_tensorflow import ViT
    from vit_tensorflow.deepvit import DeepViT
    from vit_tensorflow.mae import MAE
    from vit_tensorflow.simmim import SimMIM
    rng = np.random.default_rng(seed)
    lowp = compute == "bf16"
    q = ref_torch.bf16_round if lowp else None
    tol, gtol = (1e-4, 2e-4) if not lowp else (1.5e-2, 8e-2)
    fails = []
    t0 = time.time()
    for i in range(n):
        kind = "mae" if rng.random() < 0.5 else "simmim"
        variant = "vit" if rng.random() < 0.7 else "deepvit"
        p = int(rng.choice([4, 8]))
        g = int(rng.integers(2, 6))
        dim = 64 * int(rng.integers(1, 4)) if lowp else int(rng.choice([24, 32, 40, 64]))
        heads = int(rng.choice([1, 2, 4]))
        dh = 64 if lowp else int(rng.choice([8, 16]))
        ekw = dict(image_size=p * g, patch_size=p, num_classes=5, dim=dim, depth=int(rng.integers(1, 3)), heads=heads, mlp_dim=2 * dim, dim_head=dh)
        b = int(rng.integers(1, 5))
        npat = g * g
        ratio = float(rng.choice([1.0 / npat + 1e-3, 0.25, 0.5, 0.75, 1.0 - 1.0 / npat - 1e-3]))
        ratio = min(max(ratio, 1.0 / npat + 1e-3), 1.0 - 1.0 / npat - 1e-3)
        tag = f"#{i} {kind} {variant} b={b} ratio={ratio:.3f} {ekw}"
        try:
            ecfg = spec.make_config(variant, **ekw)
            E = spec.init_params(ecfg, 100 + i, randomize_all=True)
            enc = (ViT if variant == "vit" else DeepViT)(**ekw, compute=compute, max_batch=b, seed=0)
            enc.load_state_dict({k: np.asarray(a, np.float32) for k, a in E.items()})
            img = rng.standard_normal((b, p * g, p * g, 3)).astype(np.float32)
            wr = np.random.default_rng(200 + i)
            if kind == "mae":
                ddim = 64 * int(rng.integers(1, 3)) if lowp else int(rng.choice([16, 24, dim]))
                dkw = dict(decoder_dim=ddim, decoder_depth=1, decoder_heads=2, decoder_dim_head=32 if lowp else 8)
                w = MAE(image_size=ecfg["image_size"], encoder=enc, masking_ratio=ratio, literal_loss=False, seed=3, **dkw)
                sd = {k: (0.3 * wr.standard_normal(v.shape)).astype(np.float32) for k, v in w.state_dict().items()}
                w.load_state_dict(sd)
                dcfg = spec.make_config("vit", image_size=ecfg["image_size"], patch_size=ecfg["patch_size"], num_classes=1, dim=ddim, depth=1, heads=2,
                                        mlp_dim=4 * ddim, dim_head=dkw["decoder_dim_head"])
                D = spec.init_params(dcfg, 300 + i, randomize_all=True)
                w.decoder.load_state_dict({k: np.asarray(a, np.float32) for k, a in D.items()})
                _, nm = w.num_masked()
                idx = np.argsort(rng.uniform(size=(b, npat)), axis=-1).astype(np.int32)
                loss = w(img, indices=idx)
                grads = w.backward()
                rl, rpred, ge, gd, gw = RW.mae_forward_backward(ecfg, dcfg, E, D, {k: v.astype(np.float64) for k, v in sd.items()}, img, idx, ratio,
                                                             literal_loss=False, q=q)
                errs = {"loss": abs(loss - rl) / max(1e-30, abs(rl)), "pred": _rel(w.read("pred"), rpred)}
                for k, r in gw.items():
                    errs[k] = _rel(grads[k], r, lowp)
                for k, r in ge.items():
                    errs["encoder." + k] = _rel(grads["encoder." + k], r, lowp)
                for k, r in gd.items():
                    if k.startswith("transformer."):
                        errs["decoder." + k] = _rel(grads["decoder." + k], r, lowp)
            else:
                w = SimMIM(image_size=ecfg["image_size"], encoder=enc, masking_ratio=ratio, seed=3)
                sd = {k: (0.3 * wr.standard_normal(v.shape)).astype(np.float32) for k, v in w.state_dict().items()}
                w.load_state_dict(sd)
                _, nm = w.num_masked()
                idx = np.argsort(-rng.uniform(size=(b, npat)), axis=-1)[:, :nm].astype(np.int32)
                loss = w(img, indices=idx)
                grads = w.backward()
                rl, rpred, ge, gw = RW.simmim_forward_backward(ecfg, E, {k: v.astype(np.float64) for k, v in sd.items()}, img, idx, ratio, q=q)
                errs = {"loss": abs(loss - rl) / max(1e-30, abs(rl)), "pred": _rel(w.read("pred"), rpred)}
                for k, r in gw.items():
                    errs[k] = _rel(grads[k], r, lowp)
                for k, r in ge.items():
                    errs["encoder." + k] = _rel(grads["encoder." + k], r, lowp)
            skip = lambda k: np.asarray((gw.get(k) if k in gw else 0)).size <= 1 and lowp
            bad = {k: v for k, v in errs.items() if not (np.isfinite(v) and v <= (tol if k in ("loss", "pred") else gtol)) and not (lowp and ("reattn" in k))}
            wk = max(errs.items(), key=lambda kv: kv[1] / (tol if kv[0] in ("loss", "pred") else gtol))
            print(f"{'FAIL' if bad else 'ok  '} {tag} masked {nm}/{npat}: loss {errs['loss']:.2e}, pred {errs['pred']:.2e}, worst {wk[1]:.2e} ({wk[0]})"
                  + (f" BAD {bad}" if bad else ""), flush=True)
            if bad:
                fails.append(tag)
            del w, enc
        except Exception as ex:
            print(f"FAIL {tag}: {type(ex).__name__}: {ex}", flush=True)
            fails.append(tag)
    print(f"{n - len(fails)} / {n} wrapper configurations within the {compute} gates (values {tol:g}, gradients {gtol:g}{' relative L2' if lowp else ''}); {time.time() - t0:.0f} s")
    return fails


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0, sys.argv[3] if len(sys.argv) > 3 else "fp32") else 0)
