"""Host-side base of the masked-image-modelling wrappers (vit_tensorflow/mae.py:17, simmim.py:68): owns a `vitx_mim` handle
built around the encoder's handle, the wrapper's own parameter blob, and the Keras-like weight surface.  No arithmetic here."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import numpy as np

from . import _native as N
from ._model import VitxModel, _Weight, pair


class MimWrapper:
    _kind = N.MIM_MAE

    def _init_mim(self, image_size, encoder: VitxModel, masking_ratio: float, *, decoder_dim=0, decoder_depth=0, decoder_heads=0,
                  decoder_dim_head=0, literal_loss=True, seed=None, mpp=None):
        assert masking_ratio > 0 and masking_ratio < 1, 'masking ratio must be kept between 0 and 1'   # mae.py:28, simmim.py:71
        assert isinstance(encoder, VitxModel), "encoder must be a vit_tensorflow ViT / DeepViT"
        self.masking_ratio = masking_ratio
        self.encoder = encoder
        self.image_size = pair(image_size)
        encoder.build(input_shape=(1, self.image_size[0], self.image_size[1], 3))                      # mae.py:32, simmim.py:75
        cfg = N.MimConfig()
        cfg.kind = self._kind
        cfg.decoder_dim, cfg.decoder_depth, cfg.decoder_heads, cfg.decoder_dim_head = decoder_dim, decoder_depth, decoder_heads, decoder_dim_head
        cfg.literal_loss = 1 if literal_loss else 0
        cfg.masking_ratio = float(masking_ratio)
        if mpp is not None:                      # MPP(...) arguments (mpp.py:133-146)
            cfg.output_channel_bits = int(mpp["output_channel_bits"])
            cfg.max_pixel_val = float(mpp["max_pixel_val"])
            mean, std = mpp.get("mean"), mpp.get("std")
            if mean and std:                     # `if exists(self.mean) and exists(self.std)` (mpp.py:108), mean / std truthy (:102-103)
                assert len(mean) <= 4 and len(mean) == len(std), "mean / std: one value per channel, at most 4"
                cfg.has_norm = 1
                for i, (a, s_) in enumerate(zip(mean, std)):
                    cfg.norm_mean[i], cfg.norm_std[i] = float(a), float(s_)
        self._mcfg = cfg
        self._mim: Optional[C.c_void_p] = None
        self._enc_gen = -1
        self._rng = np.random.default_rng(seed)
        self._table: List = []
        self._n = 0
        self._blob: Optional[np.ndarray] = None
        self._device_newer = False
        self.decoder: Optional[VitxModel] = None
        self._ensure(1)

    # ---- handle management: the wrapper's device plan hangs off the encoder's handle and follows it when that is rebuilt
    def _ensure(self, batch: int):
        l = N.lib()
        enc = self.encoder
        enc_will_rebuild = enc._handle is None or batch > enc._cfg.max_batch
        stale = self._mim is not None and self._enc_gen != enc._handle_gen     # someone else made the encoder rebuild its plan
        dec_weights = None
        if self._mim is not None and (enc_will_rebuild or stale):
            # tear the wrapper's plan down BEFORE the encoder handle it points into goes away
            if not stale:
                self._pull_params()
            if self.decoder is not None:
                dec_weights = self.decoder.get_weights()
            N.check(l.vitx_mim_destroy(self._mim))
            self._mim = None
        eh = enc._ensure_handle(batch)
        if self._mim is not None:
            return self._mim
        m = C.c_void_p()
        N.check(l.vitx_mim_create(eh, C.byref(self._mcfg), C.byref(m)))
        self._mim, self._enc_gen = m, enc._handle_gen
        first = self._blob is None
        self._table, self._n = N.mim_param_table(m)
        if first:
            self._blob = np.zeros(self._n, dtype=np.float32)
            self._init_weights()
        dh = l.vitx_mim_decoder(m)
        if dh:
            self._adopt_decoder(C.c_void_p(dh), dec_weights)
        self._push_params()
        return m

    def _adopt_decoder(self, handle: C.c_void_p, weights):
        """MAE's decoder Transformer (mae.py:43) as a model object over the handle the wrapper owns."""
        from .vit import ViT
        cfg = N.Config()
        N.check(N.lib().vitx_get_config(handle, C.byref(cfg)))
        if self.decoder is None:
            dec = ViT.__new__(ViT)
            dec._init_common(image_size=(cfg.image_h, cfg.image_w), patch_size=(cfg.patch_h, cfg.patch_w), num_classes=cfg.num_classes,
                             dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, mlp_dim=cfg.mlp_dim, pool='cls', dim_head=cfg.dim_head,
                             dropout=0.0, emb_dropout=0.0, compute=self.encoder.compute, max_batch=cfg.max_batch,
                             seed=int(self._rng.integers(0, 2 ** 31 - 1)))
            self.decoder = dec
        dec = self.decoder
        dec._cfg.max_batch = cfg.max_batch
        dec._handle, dec._borrowed = handle, True
        if weights is not None:
            dec.set_weights(weights)
        else:
            dec._push_params()

    # tf.random.normal for the mask token (mae.py:42, simmim.py:83); Keras Embedding 'uniform' (+-0.05); Dense glorot_uniform / zeros
    def _init_weights(self):
        rng = self._rng
        for name, shape, off in self._table:
            n = int(np.prod(shape))
            leaf = name.split(".")[-1]
            if name == "mask_token":
                v = rng.standard_normal(n)
            elif leaf == "embeddings":
                v = rng.uniform(-0.05, 0.05, n)
            elif leaf == "kernel":
                lim = math.sqrt(6.0 / (shape[0] + shape[1]))
                v = rng.uniform(-lim, lim, n)
            else:
                v = np.zeros(n)
            self._blob[off:off + n] = v.astype(np.float32)

    def _push_params(self):
        if self._mim is not None:
            N.check(N.lib().vitx_mim_set_params(self._mim, self._blob.ctypes.data_as(C.c_void_p), self._n))
        self._device_newer = False

    def _pull_params(self):
        if self._mim is not None and self._device_newer:
            N.check(N.lib().vitx_mim_get_params(self._mim, self._blob.ctypes.data_as(C.c_void_p), self._n))
            self._device_newer = False

    def __del__(self):
        try:
            if getattr(self, "_mim", None) is not None:
                N.lib().vitx_mim_destroy(self._mim)
                self._mim = None
        except Exception:
            pass

    def _check_live(self, what: str):
        if self._mim is None or self._enc_gen != self.encoder._handle_gen:
            raise N.VitxError(N.ERR_STATE, f"{what} requires a preceding forward (the encoder's device plan was rebuilt since)")

    # ---- Keras-like surface of the wrapper's own variables
    @property
    def weights(self) -> List[_Weight]:
        return [_Weight(self, n, s, o) for n, s, o in self._table]

    def get_weights(self) -> List[np.ndarray]:
        self._pull_params()
        return [self._blob[o:o + int(np.prod(s))].reshape(s).copy() for _, s, o in self._table]

    def set_weights(self, weights) -> None:
        assert len(weights) == len(self._table), f"expected {len(self._table)} arrays, got {len(weights)}"
        for w, (n, s, o) in zip(weights, self._table):
            a = np.asarray(w, dtype=np.float32)
            # the reference's own shape, or the same values without / with singleton axes (MPP's mask token is [1, 1, c p^2], mpp.py:159)
            assert tuple(d for d in a.shape if d != 1) == tuple(d for d in s if d != 1), f"{n}: expected shape {tuple(s)}, got {a.shape}"
            self._blob[o:o + a.size] = a.reshape(-1)
        self._push_params()

    def state_dict(self) -> Dict[str, np.ndarray]:
        return {n: w for (n, _, _), w in zip(self._table, self.get_weights())}

    def load_state_dict(self, sd: Dict[str, np.ndarray]) -> None:
        self.set_weights([sd[n] for n, _, _ in self._table])

    def num_masked(self, H: Optional[int] = None, W: Optional[int] = None):
        """(num_patches, num_masked) at an image size: num_masked = int(masking_ratio * num_patches) (mae.py:57, simmim.py:106)."""
        H = H or self.image_size[0]
        W = W or self.image_size[1]
        npat, nm = C.c_int32(), C.c_int32()
        self._ensure(1)
        N.check(N.lib().vitx_mim_num_masked(self._mim, H, W, C.byref(npat), C.byref(nm)))
        return int(npat.value), int(nm.value)

    def _draw_indices(self, b: int, num_patches: int, num_masked: int) -> np.ndarray:
        raise NotImplementedError

    def __call__(self, img, training=True, indices=None, **kwargs):
        """MAE.call / SimMIM.call (mae.py:47, simmim.py:86): the reconstruction loss of one batch.  `indices` fixes the random
        masking (default: drawn like the reference does); the last draw stays readable as `.last_indices`."""
        x, _ = VitxModel._as_host(img)
        assert x.ndim == 4, "expected NHWC images [b, H, W, C]"
        b, H, W, Cc = x.shape
        m = self._ensure(b)
        npat, nm = self.num_masked(H, W)
        if indices is None:
            indices = self._draw_indices(b, npat, nm)
        idx = np.ascontiguousarray(np.asarray(indices, dtype=np.int32))
        self.last_indices = idx
        loss = np.zeros(1, dtype=np.float32)
        seed = int(kwargs.get("seed", np.random.randint(0, 2 ** 31 - 1)))   # Dropout masks of the encoder (training=training, mae.py:69)
        N.check(N.lib().vitx_mim_forward(m, x.ctypes.data_as(C.c_void_p), b, H, W, idx.ctypes.data_as(C.c_void_p),
                                         1 if training else 0, seed, loss.ctypes.data_as(C.c_void_p)))
        return loss[0]

    call = __call__

    def backward(self) -> Dict[str, np.ndarray]:
        """Gradient of the last loss w.r.t. every trainable variable (what GradientTape.gradient(loss, mae.trainable_variables)
        is meant to return, README.md:746-749): wrapper variables under their own names, the encoder's under 'encoder.<name>',
        MAE's decoder Transformer under 'decoder.<name>'."""
        self._check_live("backward")
        l = N.lib()
        self.encoder._refuse_exchange(type(self).__name__ + ".backward")
        N.check(l.vitx_mim_backward(self._mim))
        g = np.empty(self._n, dtype=np.float32)
        N.check(l.vitx_mim_get_grads(self._mim, g.ctypes.data_as(C.c_void_p), self._n))
        out = {n: g[o:o + int(np.prod(s))].reshape(s) for n, s, o in self._table}
        for prefix, model in (("encoder", self.encoder), ("decoder", self.decoder)):
            if model is None:
                continue
            mg = np.empty(model._n, dtype=np.float32)
            N.check(l.vitx_get_grads(model._handle, mg.ctypes.data_as(C.c_void_p), model._n))
            for n, s, o in model._table:
                if prefix == "decoder" and not n.startswith("transformer."):
                    continue
                out[f"{prefix}.{n}"] = mg[o:o + int(np.prod(s))].reshape(s)
        return out

    def read(self, which: str) -> np.ndarray:
        """A tensor of the last forward: 'pred', 'target', 'patches', 'encoded', 'decoded' (flat fp32)."""
        self._check_live("read")
        n = C.c_int64()
        l = N.lib()
        rc = l.vitx_mim_read(self._mim, which.encode(), np.empty(1, np.float32).ctypes.data_as(C.c_void_p), 0, C.byref(n))
        if n.value == 0:
            N.check(rc)
        out = np.empty(n.value, dtype=np.float32)
        N.check(l.vitx_mim_read(self._mim, which.encode(), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out
