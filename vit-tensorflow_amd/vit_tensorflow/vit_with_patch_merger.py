"""Drop-in for vit_tensorflow/vit_with_patch_merger.py: `ViT(..., patch_merge_layer=None, patch_merge_num_tokens=8)`
(vit_with_patch_merger.py:136-183) -- no cls token, mean pooling, and a PatchMerger (LayerNorm + learned-query attention pooling,
:42-55) after the middle layer -- on the MI355X engine."""
from ._model import VitxModel, pair  # noqa: F401
from . import _native as N


def default(val, d):
    """vit_with_patch_merger.py:13-14"""
    return val if val is not None else d


class ViT(VitxModel):
    _variant = N.VARIANT_PATCH_MERGER

    def __init__(self, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, patch_merge_layer=None, patch_merge_num_tokens=8,
                 dim_head=64, dropout=0.0, emb_dropout=0.0, **engine_kwargs):
        """Same arguments as the reference (vit_with_patch_merger.py:137-149).  Engine-only keyword extras as for vit.ViT."""
        self.patch_merge_layer_index = default(patch_merge_layer, depth // 2) - 1      # vit_with_patch_merger.py:117
        self._init_common(image_size=image_size, patch_size=patch_size, num_classes=num_classes, dim=dim, depth=depth,
                          heads=heads, mlp_dim=mlp_dim, pool='mean', dim_head=dim_head, dropout=dropout, emb_dropout=emb_dropout,
                          patch_merge_layer=patch_merge_layer, patch_merge_num_tokens=patch_merge_num_tokens, **engine_kwargs)

    @property
    def cls_token(self):
        raise AttributeError("vit_with_patch_merger.ViT has no cls_token (vit_with_patch_merger.py:163-166)")
