"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Parameter specification (names, shapes, order, initialisers) shared by the two
oracle twins (ref_numpy / ref_torch).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import anything under oracle/.

Pinned (round 2): the reference ships no tests or golden vectors and TensorFlow is not
installable here, but its source files run unmodified under oracle/tf_shim;
oracle/gen_ref_fixtures.py loads THESE parameters into the reference's own layer objects
and commits what they compute (tests/golden/ref_*.npz); tests/test_ref_fixtures.py holds
the oracle twins to those files at float64 rounding.

Shapes follow the Keras conventions the reference relies on:
  Dense kernel [in, out] (+ bias [out])              vit.py:39,42,59,63,143,156
  LayerNormalization gamma/beta [d]                  vit.py:18,155
  pos_embedding [1, Np+1, d], cls_token [1, 1, d]    vit.py:146-147
  reattn_weights [h, h]                              deepvit.py:57
  reattn_norm gamma/beta [h]                         deepvit.py:59-63
  mix_heads_pre_attn / post_attn [h, h]              cait.py:97-98
  LayerScale scale [1, 1, d]                         cait.py:43-44
  CaiT pos_embedding [1, Np, d] (no cls slot)        cait.py:168
The explicit order below is the engine's documented order (the Keras variable
order cannot be verified without TensorFlow).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np

VARIANTS = ("vit", "deepvit", "cait", "patch_merger")


def pair(t):
    """vit.py:11-12"""
    return t if isinstance(t, tuple) else (t, t)


def make_config(variant="vit", image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=6,
                heads=16, mlp_dim=2048, pool="cls", dim_head=64, cls_depth=0, num_parallel_branches=1, patch_merge_layer=None,
                patch_merge_num_tokens=8, **_ignored) -> dict:
    assert variant in VARIANTS
    ih, iw = pair(image_size)
    ph, pw = pair(patch_size)
    # vit.py:136 / deepvit.py:117 / cait.py:160
    assert ih % ph == 0 and iw % pw == 0, 'Image dimensions must be divisible by the patch size.'
    if variant not in ("cait", "patch_merger"):
        # vit.py:139 / deepvit.py:119
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
    return dict(variant=variant, image_size=(ih, iw), patch_size=(ph, pw), num_classes=num_classes,
                dim=dim, depth=depth, heads=heads, mlp_dim=mlp_dim, pool=pool, dim_head=dim_head,
                cls_depth=cls_depth, channels=3, num_parallel_branches=max(1, int(num_parallel_branches)),
                # default(patch_merge_layer, depth // 2) - 1   vit_with_patch_merger.py:117
                patch_merge_index=(patch_merge_layer if patch_merge_layer else depth // 2) - 1, patch_merge_num_tokens=patch_merge_num_tokens)


def layer_scale_init(depth_1based: int) -> float:
    """cait.py:36-41"""
    if depth_1based <= 18:
        return 0.1
    if depth_1based <= 24:
        return 1e-5
    return 1e-6


def param_spec(cfg: dict) -> List[Tuple[str, Tuple[int, ...], str]]:
    """[(name, shape, init_kind)] in the engine's explicit order.

    init_kind in {'normal', 'glorot', 'zeros', 'ones', 'const:<v>'} mirrors the reference's
    initialisers (tf.random.normal vit.py:146-147; Keras Dense glorot_uniform/zeros;
    LayerNormalization ones/zeros; LayerScale tf.fill cait.py:43).
    """
    v = cfg["variant"]
    d, h, dh, m, nc = cfg["dim"], cfg["heads"], cfg["dim_head"], cfg["mlp_dim"], cfg["num_classes"]
    inner = h * dh
    ih, iw = cfg["image_size"]
    ph, pw = cfg["patch_size"]
    np_ = (ih // ph) * (iw // pw)
    pd = ph * pw * cfg["channels"]
    spec: List[Tuple[str, Tuple[int, ...], str]] = []
    add = lambda n, s, k: spec.append((n, tuple(s), k))

    add("pos_embedding", (1, np_ if v == "cait" else np_ + 1, d), "normal")
    if v != "patch_merger":   # vit_with_patch_merger.ViT has no cls token (vit_with_patch_merger.py:163-166)
        add("cls_token", (1, 1, d), "normal")
    add("patch_embedding.kernel", (pd, d), "glorot")
    add("patch_embedding.bias", (d,), "zeros")

    def block(prefix: str, ind: int):
        if v == "cait":
            add(f"{prefix}.attn.scale", (1, 1, d), f"const:{layer_scale_init(ind + 1)}")
        add(f"{prefix}.attn.norm.gamma", (d,), "ones")
        add(f"{prefix}.attn.norm.beta", (d,), "zeros")
        if v == "cait":
            add(f"{prefix}.attn.to_q.kernel", (d, inner), "glorot")
            add(f"{prefix}.attn.to_kv.kernel", (d, 2 * inner), "glorot")
            add(f"{prefix}.attn.mix_heads_pre_attn", (h, h), "normal")
            add(f"{prefix}.attn.mix_heads_post_attn", (h, h), "normal")
        else:
            add(f"{prefix}.attn.to_qkv.kernel", (d, 3 * inner), "glorot")
        if v == "deepvit":
            add(f"{prefix}.attn.reattn_weights", (h, h), "normal")
            add(f"{prefix}.attn.reattn_norm.gamma", (h,), "ones")
            add(f"{prefix}.attn.reattn_norm.beta", (h,), "zeros")
        # vit.py:53 -- to_out disappears iff heads == 1 and dim_head == dim (ViT only)
        project_out = not (v in ("vit", "patch_merger") and h == 1 and dh == d)
        if project_out:
            add(f"{prefix}.attn.to_out.kernel", (inner, d), "glorot")
            add(f"{prefix}.attn.to_out.bias", (d,), "zeros")
        if v == "cait":
            add(f"{prefix}.mlp.scale", (1, 1, d), f"const:{layer_scale_init(ind + 1)}")
        add(f"{prefix}.mlp.norm.gamma", (d,), "ones")
        add(f"{prefix}.mlp.norm.beta", (d,), "zeros")
        add(f"{prefix}.mlp.fc1.kernel", (d, m), "glorot")
        add(f"{prefix}.mlp.fc1.bias", (m,), "zeros")
        add(f"{prefix}.mlp.fc2.kernel", (m, d), "glorot")
        add(f"{prefix}.mlp.fc2.bias", (d,), "zeros")

    P = cfg.get("num_parallel_branches", 1)
    if v == "patch_merger":   # attribute order of its Transformer: patch_merger, then the layers (vit_with_patch_merger.py:118-124)
        add("transformer.patch_merger.norm.gamma", (d,), "ones")
        add("transformer.patch_merger.norm.beta", (d,), "zeros")
        add("transformer.patch_merger.queries", (cfg["patch_merge_num_tokens"], d), "normal")
    if v == "cait":
        for i in range(cfg["depth"]):
            block(f"patch_transformer.{i}", i)
        for i in range(cfg["cls_depth"]):
            block(f"cls_transformer.{i}", i)  # LayerScale depth restarts: cait.py:142,173
    elif P > 1:
        # parallel_vit.py:104-111: layers[l] = [Parallel([PreNorm(Attention)] * P), Parallel([PreNorm(MLP)] * P)]
        assert v == "vit"
        project_out = not (h == 1 and dh == d)
        for l in range(cfg["depth"]):
            for i in range(P):
                pre = f"transformer.{l}.attn.{i}"
                add(f"{pre}.norm.gamma", (d,), "ones")
                add(f"{pre}.norm.beta", (d,), "zeros")
                add(f"{pre}.to_qkv.kernel", (d, 3 * inner), "glorot")
                if project_out:
                    add(f"{pre}.to_out.kernel", (inner, d), "glorot")
                    add(f"{pre}.to_out.bias", (d,), "zeros")
            for i in range(P):
                pre = f"transformer.{l}.mlp.{i}"
                add(f"{pre}.norm.gamma", (d,), "ones")
                add(f"{pre}.norm.beta", (d,), "zeros")
                add(f"{pre}.fc1.kernel", (d, m), "glorot")
                add(f"{pre}.fc1.bias", (m,), "zeros")
                add(f"{pre}.fc2.kernel", (m, d), "glorot")
                add(f"{pre}.fc2.bias", (d,), "zeros")
    else:
        for i in range(cfg["depth"]):
            block(f"transformer.{i}", i)

    add("mlp_head.norm.gamma", (d,), "ones")
    add("mlp_head.norm.beta", (d,), "zeros")
    add("mlp_head.kernel", (d, nc), "glorot")
    add("mlp_head.bias", (nc,), "zeros")
    return spec


def init_params(cfg: dict, seed: int = 1, randomize_all: bool = False) -> Dict[str, np.ndarray]:
    """Seeded (numpy PCG64) parameters with the reference's initialisers.

    randomize_all=True perturbs the 'zeros'/'ones'/'const' tensors too so that parity tests
    exercise biases, LN gamma/beta and LayerScale with non-trivial values.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, np.ndarray] = {}
    for name, shape, kind in param_spec(cfg):
        if kind == "normal":
            a = rng.standard_normal(shape)
        elif kind == "glorot":
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            a = rng.uniform(-lim, lim, shape)
        elif kind == "zeros":
            a = np.zeros(shape) if not randomize_all else 0.1 * rng.standard_normal(shape)
        elif kind == "ones":
            a = np.ones(shape) if not randomize_all else 1.0 + 0.1 * rng.standard_normal(shape)
        elif kind.startswith("const:"):
            c = float(kind.split(":")[1])
            a = np.full(shape, c) if not randomize_all else c * (1.0 + 0.1 * rng.standard_normal(shape))
        else:
            raise ValueError(kind)
        out[name] = np.ascontiguousarray(a, dtype=np.float64)
    return out


def flatten_params(cfg: dict, params: Dict[str, np.ndarray], dtype=np.float32) -> np.ndarray:
    return np.concatenate([np.asarray(params[n], dtype=dtype).reshape(-1) for n, _, _ in param_spec(cfg)])


def unflatten_params(cfg: dict, blob: np.ndarray) -> Dict[str, np.ndarray]:
    out, off = {}, 0
    for n, s, _ in param_spec(cfg):
        k = int(np.prod(s))
        out[n] = np.asarray(blob[off:off + k]).reshape(s)
        off += k
    assert off == blob.size
    return out


def flops_per_image(cfg: dict, fwd_only: bool = False) -> float:
    """Algorithmic FLOPs (SURVEY.md section 8d / BASELINE.md section 2): mul-add = 2, bwd = 2 x fwd."""
    d, h, dh, m, nc, L = cfg["dim"], cfg["heads"], cfg["dim_head"], cfg["mlp_dim"], cfg["num_classes"], cfg["depth"]
    inner = h * dh
    ih, iw = cfg["image_size"]
    ph, pw = cfg["patch_size"]
    np_ = (ih // ph) * (iw // pw)
    pd = ph * pw * cfg["channels"]
    n = np_ if cfg["variant"] == "cait" else np_ + 1
    L = L * cfg.get("num_parallel_branches", 1)   # parallel_vit: every layer runs P attention and P feed-forward blocks
    fwd = 2 * np_ * pd * d + L * (2 * n * d * 3 * inner + 4 * n * n * inner + 2 * n * inner * d + 4 * n * d * m) + 2 * d * nc
    return float(fwd if fwd_only else 3 * fwd)
