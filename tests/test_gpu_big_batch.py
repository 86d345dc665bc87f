"""Batches far beyond the benchmarked 256 (an MI355X holds them: 288 GB): ViT-B/16 widths at 2048 images -- 403 k token rows, the fc1 output alone is
2.48 GB, i.e. past the 31-bit byte offsets of the buffer-addressed GEMM kernels (they hand such operands to the flat-addressed form,
gemm_bf16_pipe.hip: launch_pipe).  The oracle cannot run this size; the property checked is batch independence: every image's logits are
BIT-identical to the same image run in chunks of 256 on another handle (vit.py has no cross-image operation), and the batch gradient equals the sum
of the chunks' gradients (same products, another summation tree: 1e-4 of each tensor's max)."""
import numpy as np
import pytest

from oracle import spec
from util import CONFIGS

pytestmark = pytest.mark.gpu


def test_a_2048_image_batch_equals_eight_chunks_of_256():
    from vit_tensorflow import ViT
    B, chunk = 2048, 256
    kw = dict(CONFIGS["cfg2_vit_b16"][1], depth=1)
    cfg = spec.make_config("vit", **kw)
    P = spec.init_params(cfg, 1, randomize_all=True)
    rng = np.random.default_rng(0)
    img = rng.standard_normal((B, 224, 224, 3), dtype=np.float32)
    dl = (rng.standard_normal((B, 1000), dtype=np.float32) / B).astype(np.float32)
    big = ViT(**kw, compute="bf16", max_batch=B, seed=0)
    big.load_state_dict({k: a.astype(np.float32) for k, a in P.items()})
    lb = big(img, training=False)
    gb, _ = big.backward(dl)
    assert np.isfinite(lb).all()
    del big
    small = ViT(**kw, compute="bf16", max_batch=chunk, seed=0)
    small.load_state_dict({k: a.astype(np.float32) for k, a in P.items()})
    acc = None
    for i in range(0, B, chunk):
        ls = small(img[i:i + chunk], training=False)
        assert np.array_equal(ls, lb[i:i + chunk]), f"images {i}..{i + chunk}: logits depend on the batch they ran in"
        g, _ = small.backward(dl[i:i + chunk])
        acc = {k: g[k].astype(np.float64) for k in g} if acc is None else {k: acc[k] + g[k] for k in g}
    for k in acc:
        err = np.abs(gb[k] - acc[k]).max() / (np.abs(acc[k]).max() + 1e-30)
        assert err <= 1e-4, (k, err)
