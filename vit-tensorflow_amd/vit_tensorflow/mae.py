"""Drop-in for vit_tensorflow/mae.py: `MAE(image_size, encoder, decoder_dim, masking_ratio, decoder_depth, decoder_heads,
decoder_dim_head)` and `mae(img) -> recon_loss` (mae.py:17-92) on the MI355X engine."""
import numpy as np

from . import _native as N
from ._mim import MimWrapper


class MAE(MimWrapper):
    _kind = N.MIM_MAE

    def __init__(self, image_size, encoder, decoder_dim, masking_ratio=0.75, decoder_depth=1, decoder_heads=8, decoder_dim_head=64,
                 **engine_kwargs):
        """Same arguments as the reference (mae.py:18-26).  Engine-only keyword extras: literal_loss=True keeps the loss exactly
        as mae.py:90 computes it (mean(pred**2): the second positional argument of tf.square is `name`), False gives the
        intended mean((pred - masked_patches)**2); seed=int."""
        self._init_mim(image_size, encoder, masking_ratio, decoder_dim=decoder_dim, decoder_depth=decoder_depth,
                       decoder_heads=decoder_heads, decoder_dim_head=decoder_dim_head, **engine_kwargs)

    def _draw_indices(self, b, num_patches, num_masked):
        # rand_indices = tf.argsort(tf.random.uniform([batch, num_patches]), axis=-1)   mae.py:58
        return np.argsort(self._rng.uniform(size=(b, num_patches)), axis=-1, kind="stable").astype(np.int32)
