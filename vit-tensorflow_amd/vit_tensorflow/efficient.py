"""Drop-in for vit_tensorflow/efficient.py: `ViT(image_size, patch_size, num_classes, dim, transformer, pool='cls')`
(efficient.py:12-56) -- a shell (patch embedding + cls token + position embedding in front, pooling + mlp_head behind) around a
transformer object supplied by the caller.  The shell runs in the MI355X engine (vitx_embed_forward / vitx_head_forward and their
VJPs on a depth-0 handle); the transformer is whatever the caller passes:

  * `other_model.transformer` of any ViT / DeepViT built by this package (runs in the engine, has `.backward(dout)`),
  * a `torch.nn.Module` mapping [b, n, dim] -> [b, n', dim] (differentiated with torch autograd),
  * any callable `f(tokens, training=...)`; backward then needs `f.backward(dout) -> dtokens` or `(grads, dtokens)`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._model import VitxModel, pair
from . import _native as N


class ViT(VitxModel):
    _variant = N.VARIANT_VIT

    def __init__(self, image_size, patch_size, num_classes, dim, transformer, pool='cls', **engine_kwargs):
        """Same arguments as the reference (efficient.py:13).  Engine-only keyword extras as for vit.ViT."""
        image_size_h, image_size_w = pair(image_size)
        ph, pw = pair(patch_size)
        assert image_size_h % ph == 0 and image_size_w % pw == 0, 'image dimensions must be divisible by the patch size'   # efficient.py:18
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'                  # efficient.py:19
        # depth 0: the handle holds pos_embedding, cls_token, patch_embedding.* and mlp_head.* only; heads / dim_head / mlp_dim
        # describe blocks that do not exist (any values the engine accepts)
        self._init_common(image_size=image_size, patch_size=patch_size, num_classes=num_classes, dim=dim, depth=0, heads=1,
                          mlp_dim=64, pool=pool, dim_head=64, dropout=0.0, emb_dropout=0.0, **engine_kwargs)
        self.transformer = transformer                                                                                     # efficient.py:29
        self.last_transformer_grads = None
        self._tok_in = self._tok_out = None
        self._x_shape = None

    def _is_torch_module(self) -> bool:
        try:
            import torch
        except ImportError:      # pragma: no cover
            return False
        return isinstance(self.transformer, torch.nn.Module)

    def __call__(self, img, training=True, **kwargs):
        """efficient.ViT.call (efficient.py:38-56)."""
        x, proto = self._as_host(img)
        assert x.ndim == 4, "expected NHWC images [b, H, W, C]"
        b, H, W, Cc = x.shape
        assert Cc == self._cfg.channels, f"expected {self._cfg.channels} channels"
        assert H % self._cfg.patch_h == 0 and W % self._cfg.patch_w == 0, 'image dimensions must be divisible by the patch size'
        h = self._ensure_handle(b)
        self._img_shape = (b, H, W, Cc)
        n = (H // self._cfg.patch_h) * (W // self._cfg.patch_w) + 1
        tokens = np.empty((b, n, self.dim), dtype=np.float32)
        l = N.lib()
        N.check(l.vitx_embed_forward(h, x.ctypes.data_as(C.c_void_p), b, H, W, tokens.ctypes.data_as(C.c_void_p)))     # efficient.py:40-46
        if self._is_torch_module():
            import torch
            self._tok_in = torch.from_numpy(tokens).requires_grad_(True)
            self.transformer.train(bool(training))
            self._tok_out = self.transformer(self._tok_in)                                                                # efficient.py:47
            y = self._tok_out.detach().to("cpu").float().numpy()
        else:
            self._tok_in = self._tok_out = None
            y, _ = self._as_host(self.transformer(tokens, training=training))
        assert y.ndim == 3 and y.shape[0] == b and y.shape[2] == self.dim, "the transformer must return [b, n, dim]"
        y = np.ascontiguousarray(y, dtype=np.float32)
        out = np.empty((b, self.num_classes), dtype=np.float32)
        N.check(l.vitx_head_forward(h, y.ctypes.data_as(C.c_void_p), b, y.shape[1], out.ctypes.data_as(C.c_void_p)))     # efficient.py:49-54
        self._x_shape = y.shape
        return self._like(out, proto)

    call = __call__

    def backward(self, dlogits, want_dimg: bool = False):
        """VJP of the last call: returns ({name: grad} for the shell's own parameters, dimg|None).  What the transformer reported
        for its own parameters (if anything) is kept in `last_transformer_grads`."""
        if self._handle is None or self._x_shape is None:
            raise N.VitxError(N.ERR_STATE, "backward requires a preceding forward")
        l = N.lib()
        d, _ = self._as_host(dlogits)
        dx = np.empty(self._x_shape, dtype=np.float32)
        N.check(l.vitx_head_backward(self._handle, d.ctypes.data_as(C.c_void_p), dx.ctypes.data_as(C.c_void_p)))
        self.last_transformer_grads = None
        if self._tok_out is not None:          # torch module: autograd through the caller's transformer
            import torch
            for p in self.transformer.parameters():
                p.grad = None
            self._tok_in.grad = None
            self._tok_out.backward(torch.from_numpy(dx).to(self._tok_out.device, self._tok_out.dtype))
            dtok = self._tok_in.grad.detach().float().numpy()
            self.last_transformer_grads = {k: (p.grad.detach().cpu().numpy() if p.grad is not None else None)
                                           for k, p in self.transformer.named_parameters()}
        elif hasattr(self.transformer, "backward"):
            r = self.transformer.backward(dx)
            if isinstance(r, tuple):
                self.last_transformer_grads, dtok = r
            else:
                dtok = r
            dtok, _ = self._as_host(dtok)
        else:
            raise N.VitxError(N.ERR_UNSUPPORTED, "the caller-supplied transformer has no backward(dout): cannot differentiate through it")
        dtok = np.ascontiguousarray(dtok, dtype=np.float32)
        b, H, W, Cc = self._img_shape
        assert dtok.shape == (b, (H // self._cfg.patch_h) * (W // self._cfg.patch_w) + 1, self.dim), "d(tokens) has the wrong shape"
        dimg = np.empty(self._img_shape, dtype=np.float32) if want_dimg else None
        N.check(l.vitx_embed_backward(self._handle, dtok.ctypes.data_as(C.c_void_p), dimg.ctypes.data_as(C.c_void_p) if want_dimg else None))
        g = np.empty(self._n, dtype=np.float32)
        N.check(l.vitx_get_grads(self._handle, g.ctypes.data_as(C.c_void_p), self._n))
        return {n: g[o:o + int(np.prod(s))].reshape(s) for n, s, o in self._table}, dimg
