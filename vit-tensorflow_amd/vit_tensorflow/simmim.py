"""Drop-in for vit_tensorflow/simmim.py: `SimMIM(image_size, encoder, masking_ratio)` and `mim(img) -> recon_loss`
(simmim.py:68-130) on the MI355X engine."""
import numpy as np

from . import _native as N
from ._mim import MimWrapper


class SimMIM(MimWrapper):
    _kind = N.MIM_SIMMIM

    def __init__(self, image_size, encoder, masking_ratio=0.5, **engine_kwargs):
        """Same arguments as the reference (simmim.py:69).  Engine-only keyword extra: seed=int."""
        engine_kwargs.pop("literal_loss", None)
        self._init_mim(image_size, encoder, masking_ratio, **engine_kwargs)

    def _draw_indices(self, b, num_patches, num_masked):
        # masked_indices = tf.math.top_k(tf.random.uniform([batch, num_patches]), k=num_masked).indices   simmim.py:108
        u = self._rng.uniform(size=(b, num_patches))
        return np.argsort(-u, axis=-1, kind="stable")[:, :num_masked].astype(np.int32)
