"""Drop-in for vit_tensorflow/mpp.py: `MPP(image_size, transformer, patch_size, ...)` and `mpp_trainer(images) -> loss`
(mpp.py:133-218) on the MI355X engine: the encoder's embedding, its transformer on all tokens, `to_bits` and the loss over the masked
positions are one launch sequence on the device (csrc/mim.hip, kind MPP).

Two things the reference's code does not do the way it evidently means to are reproduced LITERALLY by default, because that is what the
reference computes when it runs (tests/golden/ref_mpp_*.npz are produced by its own source under oracle/tf_shim):

  * mpp.py:185,190 write the random-patch / mask-token replacements into `masked_input.numpy()[...]`.  `EagerTensor.numpy()` returns a copy,
    so the replacements never reach `masked_input`: the transformer sees the unmasked patches and `mask_token` has no path to the loss.
    The probabilities `replace_prob` / `random_patch_prob` are accepted and stored; they influence nothing (as in the reference).
  * MPPLoss.call (mpp.py:112,125) clamps the target to [max_pixel_val, max_pixel_val] and passes (predictions, labels) to
    tf.nn.softmax_cross_entropy_with_logits(labels, logits) in swapped order: the loss as written is log(2^(bits c)) * mean_i sum_j logits_ij.
    `literal_loss=False` selects what the code evidently means: softmax cross-entropy of the masked positions' logits against the
    discretised mean colour of their patches (target clamped to [0, max_pixel_val]).
    Only the literal LOSS VALUE is pinned to the reference: its gradient here is that of the shim's broadcasting cross-entropy, which real
    TensorFlow's registered gradient (on the un-broadcast labels) may not reproduce, and the literal loss is linear in the logits -- unbounded
    below.  To TRAIN, pass literal_loss=False."""
import math

import numpy as np

from . import _native as N
from ._mim import MimWrapper


class MPP(MimWrapper):
    _kind = N.MIM_MPP

    def __init__(self, image_size, transformer, patch_size, output_channel_bits=3, channels=3, max_pixel_val=1.0, mask_prob=0.15,
                 replace_prob=0.5, random_patch_prob=0.5, mean=None, std=None, **engine_kwargs):
        """Same arguments as the reference (mpp.py:134-146).  Engine-only keyword extras: literal_loss=bool (default True), seed=int."""
        literal_loss = engine_kwargs.pop("literal_loss", True)
        assert channels == transformer._cfg.channels, "channels must match the transformer's"
        assert patch_size == transformer._cfg.patch_h == transformer._cfg.patch_w, "patch_size must be the transformer's (square) patch size"
        self.patch_size = patch_size
        self.mask_prob, self.replace_prob, self.random_patch_prob = mask_prob, replace_prob, random_patch_prob
        self._init_mim(image_size, transformer, mask_prob, literal_loss=literal_loss,
                       mpp=dict(output_channel_bits=output_channel_bits, max_pixel_val=max_pixel_val, mean=mean, std=std), **engine_kwargs)
        self.transformer = self.encoder           # the reference's attribute name (mpp.py:151)

    def num_masked(self, H=None, W=None):
        """(num_patches, max_masked) with max_masked = math.ceil(mask_prob * seq_len) (mpp.py:80)."""
        return super().num_masked(H, W)

    def _draw_indices(self, b, num_patches, num_masked):
        # rand = tf.random.uniform([batch, seq_len]); _, sampled_indices = tf.math.top_k(rand, k=max_masked)     mpp.py:82-83
        u = self._rng.uniform(size=(b, num_patches))
        assert num_masked == min(num_patches, math.ceil(self.mask_prob * num_patches))
        return np.argsort(-u, axis=-1, kind="stable")[:, :num_masked].astype(np.int32)
