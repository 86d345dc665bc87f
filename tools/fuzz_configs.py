#!/usr/bin/env python3
"""Randomised-configuration parity sweep: random ViT / DeepViT / CaiT constructor arguments (the reference's kwargs, vit.py:107-108,
deepvit.py:113-114, cait.py:150-151), random batch, forward + full backward on the GPU through the C ABI against the oracle
(oracle/ref_torch.py, fp64) on identical weights and inputs.

    python tools/fuzz_configs.py [n=40] [seed=0] [compute=fp32|bf16|bf16x3] [mode=shapes|tokens|siblings|wide|medium|sequences|sibling_sequences|medium_sequences]

mode "sequences": one handle per configuration, several calls with changing batch / image size / weights, one or two backward passes per forward.
mode "medium" (and "medium_sequences"): 256 .. 768 wide, 101 .. 257 tokens, 4 .. 24 images -- the shapes the big kernels take.
mode "wide": dim 768 .. 4096, 8 .. 32 heads, mlp_dim up to 8192 on 5 .. 17 tokens.
mode "siblings": parallel_vit.ViT (2-3 branches) and vit_with_patch_merger.ViT (random merge layer / token count).
mode "tokens": 64 .. 400 tokens per image (the dispatch boundaries of the fused attention kernels and of the 64-key sweeps of the head-axis kernels).

fp32 / bf16x3: logits <= 1e-3 abs, every gradient <= 1e-3 of its tensor's max (north_star's tolerance).  bf16: dims are drawn as multiples of 64
(the mode's requirement); gates = about twice what 80 configurations produced on MI355X (`profiles/r6/fuzz_*_r6ao.log`): logits 2e-2 of
max(1, max |logit|), gradients 9e-2 of the tensor's max -- 1.7e-1 for the [h, h] head-mix matrices (reattn_weights, mix_heads_*), whose gradient is
a sum over every score of the batch that largely cancels (with one head it is a single number).  Prints one line per configuration and a
summary; exit code 1 on any failure.  tests/test_gpu_fuzz.py runs a fixed-seed slice of it."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vit-tensorflow_amd")]


def _ungated(compute):
    """Tensors of at most this many elements are compared but NOT gated: with one head the head-mix "matrices" (cait.py:97-98, deepvit.py:57) and the
    head-axis LayerNorm parameters (deepvit.py:59-63) are single numbers -- ONE sum over every score of the batch that cancels almost completely, so
    the error relative to the tensor's own value says nothing (a float32 run of the ORACLE is 1.3 % off its float64 run on such a gradient of 1.7e-7;
    bf16 is 30-90 % off at 150-250 tokens while the same configuration is at 1e-5 in BF16X3); in bf16 the same holds for two heads (2 x 2 matrices,
    a LayerNorm over two values is +-1 whatever they are)."""
    return 4 if compute == "bf16" else 1


def draw_siblings(rng, compute):
    """parallel_vit.ViT (parallel_vit.py:119-172) and vit_with_patch_merger.ViT (vit_with_patch_merger.py:136-183): the ViT kwargs plus their own"""
    _, kw, b = draw(rng, compute, force="vit")
    kw.pop("pool", None)
    if rng.random() < 0.5:
        kw["num_parallel_branches"] = int(rng.integers(2, 4))
        kw["pool"] = "cls" if rng.random() < 0.5 else "mean"
        return "parallel_vit", kw, b
    kw["depth"] = int(rng.integers(2, 5))
    kw["patch_merge_layer"] = None if rng.random() < 0.5 else int(rng.integers(1, kw["depth"]))
    kw["patch_merge_num_tokens"] = int(rng.integers(1, 10))
    return "patch_merger", kw, b


def draw(rng, compute, force=None):
    variant = force or ["vit", "deepvit", "cait"][int(rng.integers(0, 3))]
    lowp = compute != "fp32"
    ph, pw = (int(rng.choice([4, 8, 16])), int(rng.choice([4, 8, 16])))
    if variant != "vit":
        pw = ph                                   # deepvit.py:117 / cait.py:155: int sizes only
    gh, gw = int(rng.integers(1, 7)), int(rng.integers(1, 7))
    if variant != "vit":
        gw = gh
    if variant == "cait" and gh * gw < 2:
        gh = gw = 2
    H, W = gh * ph, gw * pw
    heads = int(rng.choice([1, 2, 3, 4, 6]))
    if lowp:
        dim_head = 64 if rng.random() < 0.6 else int(rng.choice([32, 64, 128]))
        if (heads * dim_head) % 64:
            heads = 2
        dim = 64 * int(rng.integers(1, 5))
        mlp = 64 * int(rng.integers(1, 7))
    else:
        dim_head = int(rng.choice([8, 12, 16, 20, 32, 64]))
        dim = int(rng.choice([24, 36, 40, 64, 72, 100, 128]))
        mlp = int(rng.choice([20, 48, 64, 100, 130, 192]))
    kw = dict(image_size=(H, W) if variant == "vit" else H, patch_size=(ph, pw) if variant == "vit" else ph,
              num_classes=int(rng.choice([3, 10, 17, 64, 100])), dim=dim, depth=int(rng.integers(1, 4)), heads=heads, mlp_dim=mlp, dim_head=dim_head)
    if variant == "vit":
        kw["pool"] = "cls" if rng.random() < 0.6 else "mean"
        if heads == 1 and rng.random() < 0.4:
            kw["dim_head"] = dim                  # vit.py:53: to_out is the identity
            if lowp and dim % 64:
                kw["dim_head"] = 64
    if variant == "cait":
        kw["cls_depth"] = int(rng.integers(1, 3))
    b = int(rng.integers(1, 6))
    return variant, kw, b


# token counts around the dispatch boundaries of the fused attention kernels (attn_bf16.hip: 4 / 6 / 14 / 18 key tiles, the materialised path past 288)
# and of the head-axis kernels of DeepViT / CaiT (64 keys per sweep): ViT / DeepViT have np + 1 tokens, CaiT np (+ 1 in the class stage)
_GRIDS = [(7, 9), (8, 8), (5, 13), (5, 19), (8, 12), (1, 97), (6, 37), (1, 223), (14, 16), (15, 15), (7, 41), (16, 18), (17, 17), (19, 21)]


def draw_tokens(rng, compute):
    variant = ["vit", "vit", "deepvit", "cait"][int(rng.integers(0, 4))]
    gh, gw = _GRIDS[int(rng.integers(0, len(_GRIDS)))]
    if variant != "vit":
        g = [8, 9, 10, 12, 15, 16, 17][int(rng.integers(0, 7))]   # square grids only: 64 .. 289 patches
        gh = gw = g
    p = 4
    lowp = compute != "fp32"
    heads = int(rng.choice([1, 2, 3]))
    kw = dict(image_size=(gh * p, gw * p) if variant == "vit" else gh * p, patch_size=p, num_classes=10, dim=64 if lowp else int(rng.choice([24, 40, 64])),
              depth=int(rng.integers(1, 3)), heads=heads, mlp_dim=64 if lowp else 48, dim_head=64 if lowp else int(rng.choice([8, 16, 64])))
    if variant == "vit":
        kw["pool"] = "cls" if rng.random() < 0.5 else "mean"
    if variant == "cait":
        kw["cls_depth"] = 1
    return variant, kw, int(rng.integers(1, 3))


def draw_wide(rng, compute):
    """model widths up to the engine's limit (dim <= 4096; heads <= 32 for DeepViT / CaiT) on a handful of tokens: the LayerNorm row forms (d / 256 = 3 .. 16
    values per lane), the GEMM column counts, the head-axis kernels at 24 / 32 heads"""
    variant = ["vit", "deepvit", "cait"][int(rng.integers(0, 3))]
    dim = int(rng.choice([768, 1024, 1280, 1536, 2048, 2560, 3072, 4096]))
    heads = int(rng.choice([8, 12, 16, 24, 32]))
    dh = 64 if rng.random() < 0.7 else 128
    g = int(rng.integers(2, 5))
    kw = dict(image_size=8 * g, patch_size=8, num_classes=int(rng.choice([10, 1000])), dim=dim, depth=1, heads=heads,
              mlp_dim=int(rng.choice([dim, 2 * dim, min(4 * dim, 8192)])), dim_head=dh)
    if variant == "vit":
        kw["pool"] = "cls" if rng.random() < 0.5 else "mean"
    if variant == "cait":
        kw["cls_depth"] = 1
    return variant, kw, int(rng.integers(1, 4))


def draw_medium(rng, compute):
    """the shapes the big kernels take: 256 .. 768 wide, 101 .. 257 tokens, 4 .. 24 images (thousands of token rows: the pipelined persistent GEMM with
    256- / 320-row tiles and its fused epilogues, the fused attention kernels at 14 / 18 key tiles, split-K weight gradients)"""
    variant = ["vit", "vit", "deepvit", "cait"][int(rng.integers(0, 4))]
    dim = int(rng.choice([256, 384, 512, 768]))
    g = int(rng.choice([10, 13, 14, 16])) if variant == "vit" else int(rng.choice([8, 10, 12]))
    kw = dict(image_size=16 * g, patch_size=16, num_classes=int(rng.choice([10, 100, 1000])), dim=dim, depth=int(rng.integers(1, 3)), heads=dim // 64,
              mlp_dim=int(rng.choice([dim, 2 * dim, 4 * dim])), dim_head=64)
    if variant == "vit":
        kw["pool"] = "cls" if rng.random() < 0.5 else "mean"
    if variant == "cait":
        kw["cls_depth"] = 1
    return variant, kw, int(rng.integers(4, 25))


def run(n, seed, compute, mode="shapes"):
    from oracle import ref_torch, spec
    from vit_tensorflow import ViT
    from vit_tensorflow.cait import CaiT
    from vit_tensorflow.deepvit import DeepViT
    from vit_tensorflow.parallel_vit import ViT as ParallelViT
    from vit_tensorflow.vit_with_patch_merger import ViT as MergerViT
    classes = {"vit": ViT, "deepvit": DeepViT, "cait": CaiT, "parallel_vit": ParallelViT, "patch_merger": MergerViT}
    ltol, gtol, mixtol = {"fp32": (1e-3, 1e-3, 1e-3), "bf16x3": (1e-3, 1e-3, 1e-3), "bf16": (2e-2, 9e-2, 1.7e-1)}[compute]
    is_mix = lambda k: k.endswith("reattn_weights") or "mix_heads" in k
    rng = np.random.default_rng(seed)
    fails, worst_l, worst_g = [], 0.0, 0.0
    t0 = time.time()
    for i in range(n):
        variant, kw, b = {"tokens": draw_tokens, "siblings": draw_siblings, "wide": draw_wide, "medium": draw_medium}.get(mode, draw)(rng, compute)
        cfg = spec.make_config("vit" if variant == "parallel_vit" else variant, **kw)
        P = spec.init_params(cfg, 1000 + i, randomize_all=True)
        H, W = cfg["image_size"]
        img = rng.standard_normal((b, H, W, 3)).astype(np.float32)
        dl = (rng.standard_normal((b, kw["num_classes"])) / 2).astype(np.float32)
        tag = f"#{i} {variant} b={b} {kw}"
        try:
            ref_logits, ref_grads, ref_dimg = ref_torch.forward_backward(cfg, P, img, dl, want_dimg=True)
            m = classes[variant](**kw, compute=compute, max_batch=b, seed=0)
            m.load_state_dict({k: v.astype(np.float32) for k, v in P.items()})
            logits = m(img, training=False)
            want_dimg = rng.random() < 0.3
            grads, dimg = m.backward(dl, want_dimg=want_dimg)
            if want_dimg:   # d(image): the VJP of the patch unfold + projection (vit.py:142-143)
                grads = dict(grads, **{"d(img)": dimg})
                ref_grads = dict(ref_grads, **{"d(img)": ref_dimg})
            le = float(np.abs(logits - ref_logits).max())
            lscale = float(np.abs(ref_logits).max()) + 1e-30
            rel = {k: float(np.abs(grads[k] - ref_grads[k]).max() / (np.abs(ref_grads[k]).max() + 1e-30)) for k in ref_grads}
            rel = {k: v for k, v in rel.items() if np.asarray(ref_grads[k]).size > _ungated(compute)}
            ge, gk = max((v / (mixtol if is_mix(k) else gtol), k) for k, v in rel.items())   # worst in units of its gate
            ge = rel[gk]
            lerr = le if compute != "bf16" else le / max(1.0, lscale)
            ok = lerr <= ltol and all(np.isfinite(v) and v <= (mixtol if is_mix(k) else gtol) for k, v in rel.items()) and np.isfinite(le)
            worst_l, worst_g = max(worst_l, lerr), max(worst_g, ge)
            print(f"{'ok  ' if ok else 'FAIL'} {tag}: logits {le:.2e} (max |logit| {lscale:.2f}), worst gradient {ge:.2e} ({gk})", flush=True)
            if not ok:
                fails.append(tag)
            del m
        except Exception as ex:   # an error on a configuration the reference accepts is a failure too
            print(f"FAIL {tag}: {type(ex).__name__}: {ex}", flush=True)
            fails.append(tag)
    print(f"{n - len(fails)} / {n} configurations within the {compute} gates (logits {ltol:g}, gradients {gtol:g}, head-mix matrices {mixtol:g}); worst logits {worst_l:.2e}, "
          f"worst gradient (nearest its gate) {worst_g:.2e}; "
          f"{time.time() - t0:.0f} s")
    for f in fails:
        print("failed:", f)
    return fails


def run_sequences(n, seed, compute, steps=5, siblings=False, medium=False):
    """Stale-state hunt: ONE handle per random configuration, then `steps` calls with a random batch (<= the handle's plan), a random SMALLER image
    (vit.py:165: pos_embedding[:, :n + 1] -- fewer patches than image_size are legal), new weights every now and then (the bf16 operand refresh),
    and sometimes two backward passes on one forward; every call against the oracle."""
    from oracle import ref_torch, spec
    from vit_tensorflow import ViT
    from vit_tensorflow.cait import CaiT
    from vit_tensorflow.deepvit import DeepViT
    from vit_tensorflow.parallel_vit import ViT as ParallelViT
    from vit_tensorflow.vit_with_patch_merger import ViT as MergerViT
    classes = {"vit": ViT, "deepvit": DeepViT, "cait": CaiT, "parallel_vit": ParallelViT, "patch_merger": MergerViT}
    ltol, gtol, mixtol = {"fp32": (1e-3, 1e-3, 1e-3), "bf16x3": (1e-3, 1e-3, 1e-3), "bf16": (2e-2, 9e-2, 1.7e-1)}[compute]
    is_mix = lambda k: k.endswith("reattn_weights") or "mix_heads" in k
    rng = np.random.default_rng(seed)
    fails, calls = [], 0
    t0 = time.time()
    for i in range(n):
        variant, kw, bmax = (draw_medium if medium else draw_siblings if siblings else draw)(rng, compute)
        bmax = max(bmax, 5) if medium else 5
        cfg = spec.make_config("vit" if variant == "parallel_vit" else variant, **kw)
        (H, W), (ph, pw) = cfg["image_size"], cfg["patch_size"]
        P = spec.init_params(cfg, 2000 + i, randomize_all=True)
        m = classes[variant](**kw, compute=compute, max_batch=bmax, seed=0)
        m.load_state_dict({k: v.astype(np.float32) for k, v in P.items()})
        for st in range(steps):
            b = int(rng.integers(1, bmax + 1))
            gh, gw = H // ph, W // pw
            if rng.random() < 0.4:
                gh, gw = int(rng.integers(1, gh + 1)), int(rng.integers(1, gw + 1))
                if variant in ("deepvit", "cait"):
                    gw = gh = min(gh, gw)
                if variant == "cait" and gh * gw < 2:
                    gh, gw = H // ph, W // pw
            if rng.random() < 0.3:
                P = spec.init_params(cfg, 3000 + 17 * i + st, randomize_all=True)
                m.load_state_dict({k: v.astype(np.float32) for k, v in P.items()})
            img = rng.standard_normal((b, gh * ph, gw * pw, 3)).astype(np.float32)
            tag = f"#{i}.{st} {variant} b={b} image {gh * ph}x{gw * pw} of {kw}"
            try:
                logits = m(img, training=False)
                nback = 0 if rng.random() < 0.1 else (2 if rng.random() < 0.25 else 1)
                ok, msg = True, ""
                ref_logits = None
                for _ in range(max(1, nback)):
                    dl = (rng.standard_normal((b, kw["num_classes"])) / 2).astype(np.float32)
                    ref_logits, ref_grads, _ = ref_torch.forward_backward(cfg, P, img, dl)
                    if nback == 0:
                        break
                    grads, _g = m.backward(dl)
                    rel = {k: float(np.abs(grads[k] - ref_grads[k]).max() / (np.abs(ref_grads[k]).max() + 1e-30)) for k in ref_grads}
                    rel = {k: v for k, v in rel.items() if np.asarray(ref_grads[k]).size > _ungated(compute)}
                    bad = {k: v for k, v in rel.items() if not (np.isfinite(v) and v <= (mixtol if is_mix(k) else gtol))}
                    if bad:
                        ok, msg = False, f"gradients {bad}"
                    if rng.random() < 0.3:   # an optimizer step inside the library: the next call must see the new weights (bf16 operand refresh)
                        m.apply_gradients("sgd" if rng.random() < 0.5 else "adamw", lr=3e-3)
                        P = {k: np.asarray(v, np.float64) for k, v in m.state_dict().items()}
                        break
                le = float(np.abs(logits - ref_logits).max())
                lerr = le if compute != "bf16" else le / max(1.0, float(np.abs(ref_logits).max()))
                if not (np.isfinite(le) and lerr <= ltol):
                    ok, msg = False, msg + f" logits {le:.2e}"
                calls += 1
                if not ok:
                    print(f"FAIL {tag}: {msg}", flush=True)
                    fails.append(tag)
            except Exception as ex:
                print(f"FAIL {tag}: {type(ex).__name__}: {ex}", flush=True)
                fails.append(tag)
        del m
    print(f"{calls - len(fails)} / {calls} calls on {n} handles within the {compute} gates; {time.time() - t0:.0f} s")
    return fails


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[4] in ("sequences", "sibling_sequences", "medium_sequences"):
        sys.exit(1 if run_sequences(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], siblings=sys.argv[4] == "sibling_sequences",
                                    medium=sys.argv[4] == "medium_sequences", steps=3 if sys.argv[4] == "medium_sequences" else 5) else 0)
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                      sys.argv[3] if len(sys.argv) > 3 else "fp32", sys.argv[4] if len(sys.argv) > 4 else "shapes") else 0)
