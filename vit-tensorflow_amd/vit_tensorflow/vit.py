"""Drop-in for vit_tensorflow/vit.py: same `ViT(...)` constructor and `__call__(img, training=True)`
(vit.py:106-177); the forward/backward arithmetic runs in hand-written gfx950 HIP kernels."""
from ._model import VitxModel, pair  # noqa: F401
from . import _native as N


class ViT(VitxModel):
    _variant = N.VARIANT_VIT

    def __init__(self, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim,
                 pool='cls', dim_head=64, dropout=0.0, emb_dropout=0.0, **engine_kwargs):
        """Same arguments as the reference (vit.py:107-108).  Engine-only keyword extras:
        compute='fp32'|'bf16', max_batch=int, device=int, seed=int."""
        self._init_common(image_size=image_size, patch_size=patch_size, num_classes=num_classes, dim=dim, depth=depth,
                          heads=heads, mlp_dim=mlp_dim, pool=pool, dim_head=dim_head, dropout=dropout,
                          emb_dropout=emb_dropout, **engine_kwargs)
