"""Drop-in for vit_tensorflow/parallel_vit.py: `ViT(..., num_parallel_branches=2)` (parallel_vit.py:119-172) -- every layer is a sum
of parallel attention blocks followed by a sum of parallel feed-forward blocks -- on the MI355X engine (the ordinary ViT kernels,
wired as half-blocks that normalise the layer's input and add into one running residual)."""
from ._model import VitxModel, pair  # noqa: F401
from . import _native as N


class ViT(VitxModel):
    _variant = N.VARIANT_VIT

    def __init__(self, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool='cls', num_parallel_branches=2,
                 dim_head=64, dropout=0.0, emb_dropout=0.0, **engine_kwargs):
        """Same arguments as the reference (parallel_vit.py:120-133).  Engine-only keyword extras as for vit.ViT."""
        assert 1 <= int(num_parallel_branches) <= 8, "num_parallel_branches must be in [1, 8]"
        self.num_parallel_branches = int(num_parallel_branches)
        self._init_common(image_size=image_size, patch_size=patch_size, num_classes=num_classes, dim=dim, depth=depth,
                          heads=heads, mlp_dim=mlp_dim, pool=pool, dim_head=dim_head, dropout=dropout,
                          emb_dropout=emb_dropout, num_parallel_branches=num_parallel_branches, **engine_kwargs)
