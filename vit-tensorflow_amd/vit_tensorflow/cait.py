"""Drop-in for vit_tensorflow/cait.py (CaiT: LayerScale + talking-heads attention + class-attention stage,
cait.py:33-194)."""
from ._model import VitxModel
from . import _native as N


class CaiT(VitxModel):
    _variant = N.VARIANT_CAIT

    def __init__(self, image_size, patch_size, num_classes, dim, depth, cls_depth, heads, mlp_dim,
                 dim_head=64, dropout=0.0, emb_dropout=0.0, layer_dropout=0.0, **engine_kwargs):
        """Same arguments as the reference (cait.py:156-157)."""
        self._init_common(image_size=image_size, patch_size=patch_size, num_classes=num_classes, dim=dim, depth=depth,
                          cls_depth=cls_depth, heads=heads, mlp_dim=mlp_dim, pool='cls', dim_head=dim_head,
                          dropout=dropout, emb_dropout=emb_dropout, layer_dropout=layer_dropout, **engine_kwargs)
