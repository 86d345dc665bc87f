"""The tokenizer step of vit_tensorflow/t2t.py (RearrangeUnfoldTransformer.call, t2t.py:39-47): tokens -> image grid ->
tf.image.extract_patches(sizes k, strides s, rates 1, padding 'SAME') -> tokens, as a HIP kernel behind vitx_extract_patches
(pure index arithmetic: bit-exact) with its VJP.  T2TViT itself is not provided: its inner transformers have widths 3*49 = 147
and 147*9 = 1323 (t2t.py:62-70), which the engine's 4-wide row kernels do not take."""
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
