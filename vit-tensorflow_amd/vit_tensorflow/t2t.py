"""Drop-in for vit_tensorflow/t2t.py.  The tokenizer step (RearrangeUnfoldTransformer.call, t2t.py:39-47): tokens -> image grid ->
tf.image.extract_patches(sizes k, strides s, rates 1, padding 'SAME') -> tokens, as a HIP kernel behind vitx_extract_patches
(pure index arithmetic: bit-exact) with its VJP; and T2TViT (t2t.py:49-122) composed from it, one engine handle per tokenizer
transformer (widths 3*49 = 147 and 147*9 = 1323 at the defaults: the engine's any-width LayerNorm / epilogue paths) and an ordinary
handle for the Dense + cls / position rows + transformer + head (vitx_forward_patches)."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _native as N


def conv_output_size(image_size, kernel_size, stride, padding):
    """t2t.py:15-16"""
    return int(((image_size - kernel_size + (2 * padding)) / stride) + 1)


def _host(x):
    is_torch = type(x).__module__.startswith("torch")
    if is_torch:
        return np.ascontiguousarray(x.detach().to("cpu").float().numpy()), x
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32)), None


def _like(out, proto):
    if proto is None:
        return out
    import torch
    return torch.from_numpy(out).to(proto.device)


def extract_patches_shape(H: int, W: int, Cc: int, kernel_size: int, stride: int):
    oh, ow, f = C.c_int32(), C.c_int32(), C.c_int32()
    N.check(N.lib().vitx_extract_patches_shape(H, W, Cc, kernel_size, stride, C.byref(oh), C.byref(ow), C.byref(f)))
    return oh.value, ow.value, f.value


def extract_patches(x, kernel_size: int, stride: int):
    """tf.image.extract_patches(x, [1,k,k,1], [1,s,s,1], [1,1,1,1], 'SAME') on NHWC x (t2t.py:42) -> [b, oh, ow, k*k*C]."""
    a, proto = _host(x)
    assert a.ndim == 4, "expected NHWC [b, H, W, C]"
    b, H, W, Cc = a.shape
    oh, ow, f = extract_patches_shape(H, W, Cc, kernel_size, stride)
    out = np.empty((b, oh, ow, f), dtype=np.float32)
    N.check(N.lib().vitx_extract_patches(a.ctypes.data_as(C.c_void_p), b, H, W, Cc, kernel_size, stride, out.ctypes.data_as(C.c_void_p)))
    return _like(out, proto)


def extract_patches_backward(dout, input_shape, kernel_size: int, stride: int):
    """VJP of extract_patches: d(out) [b, oh, ow, k*k*C] -> d(x) [b, H, W, C]."""
    a, proto = _host(dout)
    b, H, W, Cc = (int(v) for v in input_shape)
    oh, ow, f = extract_patches_shape(H, W, Cc, kernel_size, stride)
    assert a.shape == (b, oh, ow, f), f"expected d(out) of shape {(b, oh, ow, f)}"
    dx = np.empty((b, H, W, Cc), dtype=np.float32)
    N.check(N.lib().vitx_extract_patches_backward(a.ctypes.data_as(C.c_void_p), b, H, W, Cc, kernel_size, stride, dx.ctypes.data_as(C.c_void_p)))
    return _like(dx, proto)


class RearrangeUnfold:
    """The non-transformer part of RearrangeUnfoldTransformer (t2t.py:18-47): `is_first` takes the NHWC image, later layers take
    tokens [b, h*w, c] of a square grid (t2t.py:40-41); returns tokens [b, oh*ow, k*k*c] (t2t.py:42-43)."""

    def __init__(self, is_first: bool, kernel_size: int, stride: int):
        self.is_first, self.kernel_size, self.stride = is_first, kernel_size, stride
        self._in_shape = None
        self._tokens_in = False

    def __call__(self, x, training=True):
        a, proto = _host(x)
        self._tokens_in = not self.is_first
        if not self.is_first:
            b, n, c = a.shape
            h = int(math.sqrt(n))                                   # t2t.py:41
            assert h * h == n, "tokens must form a square grid"
            a = a.reshape(b, h, n // h, c)
        self._in_shape = a.shape
        y = extract_patches(a, self.kernel_size, self.stride)      # t2t.py:42
        return _like(y.reshape(y.shape[0], y.shape[1] * y.shape[2], y.shape[3]), proto)   # t2t.py:43

    def backward(self, dout):
        a, proto = _host(dout)
        b, H, W, Cc = self._in_shape
        oh, ow, f = extract_patches_shape(H, W, Cc, self.kernel_size, self.stride)
        dx = extract_patches_backward(a.reshape(b, oh, ow, f), self._in_shape, self.kernel_size, self.stride)
        if self._tokens_in:
            dx = dx.reshape(b, H * W, Cc)
        return _like(dx, proto)


def exists(val):
    return val is not None


class _RenamedWeight:
    """A weight handle of one of the composed engine handles under its T2TViT name."""

    def __init__(self, w, name):
        self._w, self.name = w, name

    @property
    def shape(self):
        return self._w.shape

    def numpy(self):
        return self._w.numpy()

    def assign(self, value):
        self._w.assign(value)

    def __array__(self, dtype=None):
        return self._w.__array__(dtype)

    def __getitem__(self, idx):
        return self._w[idx]


class T2TViT:
    """Drop-in for vit_tensorflow/t2t.py:49-122.  Same constructor and `model(img, training=True)`.

    The token-to-token patch embedding (t2t.py:59-75) runs as the reference composes it: per layer the unfold tokenizer
    (vitx_extract_patches, bit-exact) and -- for all but the last layer -- a one-block Transformer(dim=layer_dim, heads=1,
    dim_head=layer_dim, mlp_dim=layer_dim) on its own engine handle (widths 3*7*7 = 147 and 147*3*3 = 1323 at the defaults: the
    any-width LayerNorm / epilogue paths, fp32); the final Dense, cls / position rows, the transformer and the head are an ordinary
    engine handle fed through vitx_forward_patches.  `backward` chains the VJPs in the reverse order.  Engine-only keyword extras:
    compute='fp32'|'bf16' (the main model; the tokenizer's transformers stay fp32), max_batch, device, seed."""

    def __init__(self, image_size, num_classes, dim, depth=None, heads=None, mlp_dim=None, pool='cls', channels=3, dim_head=64, dropout=0.0,
                 emb_dropout=0.0, transformer=None, t2t_layers=((7, 4), (3, 2), (3, 2)), compute="fp32", max_batch=None, device=0, seed=None):
        from ._model import VitxModel
        from .vit import ViT
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'      # t2t.py:54
        if exists(transformer):
            raise NotImplementedError("T2TViT(transformer=...): a caller-supplied transformer object -- use vit_tensorflow.efficient.ViT for that shell")
        assert all([exists(depth), exists(heads), exists(mlp_dim)]), 'depth, heads, and mlp_dim must be supplied'   # t2t.py:83
        self.t2t_layers = tuple(tuple(x) for x in t2t_layers)
        self.pool, self.dim, self.num_classes = pool, dim, num_classes
        self.kwargs = dict(dim=dim, num_classes=num_classes)
        rng = np.random.default_rng(seed)
        self._unfold, self._inner, self.layer_dims = [], [], []
        layer_dim, out_size, grid = channels, image_size, image_size
        L = len(self.t2t_layers)
        for i, (k, s) in enumerate(self.t2t_layers):                                                             # t2t.py:60-72
            layer_dim *= k ** 2
            out_size = conv_output_size(out_size, k, s, s // 2)
            grid = -(-grid // s)                                   # token grid the SAME-padded unfold produces (ceil)
            self.layer_dims.append(layer_dim)
            self._unfold.append(RearrangeUnfold(i == 0, k, s))
            if i < L - 1:
                self._inner.append(ViT(image_size=grid, patch_size=1, num_classes=1, dim=layer_dim, depth=1, heads=1, mlp_dim=layer_dim,
                                       dim_head=layer_dim, dropout=dropout, compute="fp32", max_batch=max_batch, device=device,
                                       seed=int(rng.integers(0, 2 ** 31 - 1))))
        num_pos = out_size ** 2                                                                                   # t2t.py:77
        assert grid * grid <= num_pos, "the tokenizer produces more tokens than pos_embedding has rows"
        main = VitxModel()
        main._variant = N.VARIANT_VIT
        # the main handle sees the tokenizer's output as `num_pos` patches of layer_dim features (patch = 1 x layer_dim "pixels", 1 channel)
        main._init_common(image_size=(num_pos, layer_dim), patch_size=(1, layer_dim), num_classes=num_classes, dim=dim, depth=depth, heads=heads,
                          mlp_dim=mlp_dim, pool=pool, dim_head=dim_head, dropout=dropout, emb_dropout=emb_dropout, compute=compute,
                          max_batch=max_batch, device=device, seed=int(rng.integers(0, 2 ** 31 - 1)), channels=1)
        self._main = main
        self.transformer = main.transformer
        self.mlp_head = main.mlp_head
        self.dropout = main.dropout
        self._seeds = None

    # ---- names: tokenizer transformers, then the Dense (index L in the reference's Sequential), then the ViT part
    def _name_map(self):
        L = len(self.t2t_layers)
        out = []
        for i, m in enumerate(self._inner):
            for n, s, _ in m._table:
                if n.startswith("transformer.0."):
                    out.append((f"patch_embedding.{i}.transformer_layer.0." + n[len("transformer.0."):], m, n, tuple(s)))
        for n, s, _ in self._main._table:
            nn = {"patch_embedding.kernel": f"patch_embedding.{L}.kernel", "patch_embedding.bias": f"patch_embedding.{L}.bias"}.get(n, n)
            out.append((nn, self._main, n, tuple(s)))
        # Keras' Model.weights order: the model's own tf.Variables first (pos_embedding, cls_token: t2t.py:77-78), then the sublayers in
        # attribute order -- patch_embedding.{i}.*, patch_embedding.{L}.kernel / .bias, transformer.*, mlp_head.* (as the engine's ViT table)
        own = [e for e in out if e[0] in ("pos_embedding", "cls_token")]
        pe = [e for e in out if e[0].startswith("patch_embedding.")]
        return own + pe + [e for e in out if e[0] not in ("pos_embedding", "cls_token") and not e[0].startswith("patch_embedding.")]

    def state_dict(self):
        sds = {id(m): m.state_dict() for m in self._inner + [self._main]}
        return {nn: sds[id(m)][n] for nn, m, n, _ in self._name_map()}

    def load_state_dict(self, sd):
        for m in self._inner + [self._main]:
            cur = m.state_dict()
            for nn, mm, n, s in self._name_map():
                if mm is m:
                    a = np.asarray(sd[nn], dtype=np.float32)
                    assert a.shape == s, f"{nn}: expected shape {s}, got {a.shape}"
                    cur[n] = a
            m.load_state_dict(cur)

    def get_weights(self):
        return list(self.state_dict().values())

    def set_weights(self, weights):
        names = [e[0] for e in self._name_map()]
        assert len(weights) == len(names)
        self.load_state_dict(dict(zip(names, weights)))

    def count_params(self):
        return int(sum(int(np.prod(s)) for _, _, _, s in self._name_map()))

    @property
    def weights(self):
        """Keras-order list of weight handles (name / shape / numpy()), spanning the tokenizer's handles and the main one."""
        per = {id(m): {w.name: w for w in m.weights} for m in self._inner + [self._main]}
        out = []
        for nn, m, n, _ in self._name_map():
            w = per[id(m)][n]
            out.append(_RenamedWeight(w, nn))
        return out

    trainable_variables = weights
    trainable_weights = weights

    @staticmethod
    def _npz_path(path):
        return path if str(path).endswith(".npz") else str(path) + ".npz"

    def save_weights(self, path):
        """Weights by name in one .npz (same format and caveat as the ViT classes' save_weights)."""
        np.savez(self._npz_path(path), **self.state_dict())

    def load_weights(self, path):
        with np.load(self._npz_path(path)) as z:
            self.load_state_dict({k: z[k] for k in z.files})

    @property
    def pos_embedding(self):
        return self._main.pos_embedding

    @property
    def cls_token(self):
        return self._main.cls_token

    def patch_embedding(self, img, training=True, seed=None):
        """The token-to-token Sequential including its final Dense is only available fused with what follows; this returns the
        tokenizer's tokens [b, n, last layer_dim] (the input of that Dense)."""
        return self._tokenize(img, training, seed)

    def _tokenize(self, img, training, seed):
        seed = int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed)
        x = img
        for i, u in enumerate(self._unfold):
            x = u(x)
            if i < len(self._inner):
                x = self._inner[i].transformer(x, training=training, seed=seed + 7919 * (i + 1))   # RearrangeUnfoldTransformer.call  t2t.py:44-45
        self._last_seed = seed
        return x

    def __call__(self, img, training=True, **kwargs):
        """T2TViT.call (t2t.py:99-121)"""
        x, proto = _host(img)
        assert x.ndim == 4, "expected NHWC images [b, H, W, C]"
        tok = self._tokenize(x, training, kwargs.get("seed"))
        return _like(np.asarray(self._main.forward_patches(tok, training=training, seed=self._last_seed)), proto)

    call = __call__

    def backward(self, dlogits, want_dimg: bool = False):
        """VJP of the last call: ({name: grad}, dimg | None)."""
        grads_main, d = self._main.backward(dlogits, want_dimg=True)        # d = d(tokenizer output) [b, n, last layer_dim]
        return self._chain_tokenizer(grads_main, d, want_dimg)

    def _chain_tokenizer(self, grads_main, d, want_dimg):
        """d = d(tokenizer output) -> the tokenizer transformers' gradients (and d(img)); merges them with the main handle's."""
        inner_grads = [None] * len(self._inner)
        for i in range(len(self._unfold) - 1, -1, -1):
            if i < len(self._inner):
                inner_grads[i], d = self._inner[i].transformer.backward(d)
            if i > 0 or want_dimg:
                d = self._unfold[i].backward(d)
        out = {}
        for nn, m, n, _ in self._name_map():
            out[nn] = grads_main[n] if m is self._main else inner_grads[self._inner.index(m)][n]
        return out, (d if want_dimg else None)

    def apply_gradients(self, *args, **kw):
        for m in self._inner + [self._main]:
            m.apply_gradients(*args, **kw)
