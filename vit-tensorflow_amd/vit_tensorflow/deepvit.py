"""Drop-in for vit_tensorflow/deepvit.py (DeepViT with Re-attention, deepvit.py:46-157)."""
from ._model import VitxModel
from . import _native as N


class DeepViT(VitxModel):
    _variant = N.VARIANT_DEEPVIT

    def __init__(self, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim,
                 pool='cls', dim_head=64, dropout=0.0, emb_dropout=0.0, **engine_kwargs):
        """Same arguments as the reference (deepvit.py:113-114; image_size / patch_size are ints there)."""
        self._init_common(image_size=image_size, patch_size=patch_size, num_classes=num_classes, dim=dim, depth=depth,
                          heads=heads, mlp_dim=mlp_dim, pool=pool, dim_head=dim_head, dropout=dropout,
                          emb_dropout=emb_dropout, **engine_kwargs)
