"""vit_tensorflow -- MI355X-native engine behind the reference's import surface.

The reference ships no __init__.py although its README does `from vit_tensorflow import ViT`
(README.md:47); this package supplies it, plus `vit_tensorflow.deepvit.DeepViT` (README.md:148) and
`vit_tensorflow.cait.CaiT` (README.md:177), and the masked-image-modelling wrappers
`vit_tensorflow.mae.MAE` (README.md:685,697) / `vit_tensorflow.simmim.SimMIM` (README.md:644,656) / `vit_tensorflow.mpp.MPP` (README.md:720,734), and the distillation
pair `vit_tensorflow.distill.DistillableViT` / `DistillWrapper` (distill.py:46,87).  Sibling modules: `parallel_vit`,
`vit_with_patch_merger`, `efficient` (shell around a caller-supplied transformer, efficient.py:12) and `t2t` (the tokenizer step, t2t.py:39-47).
"""
from .vit import ViT
from .deepvit import DeepViT
from .cait import CaiT
from .mae import MAE
from .simmim import SimMIM
from .mpp import MPP
from .distill import DistillableViT, DistillWrapper

__all__ = ["ViT", "DeepViT", "CaiT", "MAE", "SimMIM", "MPP", "DistillableViT", "DistillWrapper"]
