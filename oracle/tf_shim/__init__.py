"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

A minimal stand-in for the `tensorflow` / `tensorflow.keras` names that the reference's hot-path modules import
(vit.py:1-9, deepvit.py:1-9, cait.py:1-12, parallel_vit.py:1-8, vit_with_patch_merger.py:1-8, efficient.py:1-8,
distill.py:1-11, mae.py:1-8, simmim.py:1-7, t2t.py:1-9), so that the reference's OWN SOURCE FILES under
/root/reference/vit_tensorflow/ execute unmodified in this container, where TensorFlow cannot be installed.

    import oracle.tf_shim as shim
    shim.install()                                  # registers sys.modules['tensorflow', 'tensorflow.keras', ...]
    sys.path.insert(0, '/root/reference/vit_tensorflow')
    import vit                                      # the reference's file, byte for byte
    m = vit.ViT(image_size=64, patch_size=16, ...)  # its classes, its call graph, its op order

What this pins and what it does not.  Everything the reference WRITES -- op order, split order, einsum index strings,
which tensor is normalised, where the cls token goes, residual placement, LayerScale depth thresholds, the rearrange
patterns (executed by the real `einops`, torch backend) -- runs from the reference's source.  What the reference merely
CALLS into Keras is implemented here from Keras' documented behaviour, in float64 on torch (autograd plays the role of
tf.GradientTape):

  Dense(units, use_bias=True)            y = x @ kernel[in, units] + bias; kernel glorot_uniform, bias zeros; built lazily
  LayerNormalization(axis=-1, epsilon=1e-3, center=True, scale=True)    biased variance; gamma ones, beta zeros
  Softmax(axis=-1), Activation(fn), Dropout(rate) (inverted; identity unless training and rate > 0), Embedding
  Sequential(layers, name)               forwards `training` only to layers whose call() accepts it
  Layer.__call__                         builds on first use; drops a `training` kwarg that call() does not accept
  tf.Variable / tf.random.normal / tf.fill / tf.split / tf.concat / tf.einsum / tf.matmul / tf.transpose / tf.reduce_* /
  tf.math.erf / tf.nn.softmax / tf.nn.log_softmax / tf.image.extract_patches (SAME) / the two Keras losses distill.py uses

Tensors are plain `torch.Tensor` (float64), so `einops` dispatches to its torch backend; `tf.Tensor` / `tf.Variable` are
never instances of anything (einops' TensorFlow backend therefore never claims a tensor).  `.numpy()` on a tensor that
is part of the autograd graph detaches, as leaving the TensorFlow graph does (mae.py:62, simmim.py:119).

Only oracle/gen_ref_fixtures.py and tests/ use this module; the product never imports it.
"""
from __future__ import annotations

import inspect
import math
import sys
import types
from typing import List, Optional, Sequence

import numpy as np
import torch

DTYPE = torch.float64
_GEN = torch.Generator().manual_seed(0)


def seed(s: int) -> None:
    """Seed the shim's stand-in for TensorFlow's global RNG (tf.random.*, glorot_uniform, Dropout masks)."""
    _GEN.manual_seed(int(s))


class _T(torch.Tensor):
    """float64 torch tensor whose .numpy() leaves the autograd graph the way EagerTensor.numpy() leaves the tape."""

    def numpy(self):  # noqa: D401
        # EagerTensor.numpy() returns a COPY ("Copy of the contents of this Tensor into a NumPy array"): writing into the result never reaches
        # the tensor -- mpp.py:185,190 assign into `masked_input.numpy()[...]`, which therefore leaves masked_input as it was
        return torch.Tensor.numpy(self.detach().as_subclass(torch.Tensor)).copy()

    @property
    def ndim_(self):
        return self.dim()

    # EagerTensor (op) ndarray is a tensor on the tape (simmim.py:128: pred - patches.numpy()[...]); torch would hand the pair to
    # NumPy, which detaches.  The ndarray operand becomes a constant tensor instead.
    def _lift(self, other):
        return _t(other) if isinstance(other, np.ndarray) else other

    def __add__(self, o): return torch.Tensor.__add__(self, self._lift(o))
    def __radd__(self, o): return torch.Tensor.__radd__(self, self._lift(o))
    def __sub__(self, o): return torch.Tensor.__sub__(self, self._lift(o))
    def __rsub__(self, o): return torch.Tensor.__rsub__(self, self._lift(o))
    def __mul__(self, o): return torch.Tensor.__mul__(self, self._lift(o))
    def __rmul__(self, o): return torch.Tensor.__rmul__(self, self._lift(o))
    def __truediv__(self, o): return torch.Tensor.__truediv__(self, self._lift(o))
    def __rtruediv__(self, o): return torch.Tensor.__rtruediv__(self, self._lift(o))
    __array_priority__ = 1000          # ndarray (op) tensor defers to the tensor as well


def _t(x, dtype=None) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        out = x if dtype is None else x.to(dtype)
    else:
        a = np.asarray(x)
        if dtype is None:
            dtype = DTYPE if a.dtype.kind == "f" else (torch.bool if a.dtype.kind == "b" else torch.int64)
        out = torch.as_tensor(a).to(dtype)
    return out.as_subclass(_T) if not isinstance(out, _T) else out


# ----------------------------------------------------------------------------------------------- tf.*
class Tensor:            # never instantiated: keeps einops' TensorflowBackend.is_appropriate_type() False
    pass


class Variable:
    """tf.Variable(initial_value) -> a float64 leaf tensor with requires_grad (assign with `assign(var, value)`)."""

    def __new__(cls, initial_value=None, trainable=True, name=None, dtype=None, **_):
        v = _t(initial_value).detach().clone().as_subclass(_T)
        if v.dtype.is_floating_point:
            v.requires_grad_(bool(trainable))
        return v


def assign(var: torch.Tensor, value) -> None:
    with torch.no_grad():
        var.copy_(_t(value).reshape(var.shape))


def _shape_list(shape) -> List[int]:
    return [int(s) for s in shape]


def _random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    return _t(torch.randn(_shape_list(shape), generator=_GEN, dtype=DTYPE) * stddev + mean)


def _random_uniform(shape, minval=0.0, maxval=1.0, dtype=None, seed=None, name=None):
    if dtype in (torch.int32, torch.int64):      # integers in [minval, maxval)  (mpp.py:182)
        return _t(torch.randint(int(minval), int(maxval), _shape_list(shape), generator=_GEN))
    return _t(torch.rand(_shape_list(shape), generator=_GEN, dtype=DTYPE) * (maxval - minval) + minval)


def zeros(shape, dtype=None, name=None):
    return _t(torch.zeros(_shape_list(shape), dtype=DTYPE))


def expand_dims(input, axis, name=None):
    return _t(input).unsqueeze(axis)


def reshape(tensor, shape, name=None):
    return _t(tensor).reshape(_shape_list(shape))


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    """tf.clip_by_value: min(max(t, lo), hi) -- with lo > hi every element becomes hi (mpp.py:112 passes lo = reduce_min(max_pixel_val))."""
    return torch.minimum(torch.maximum(_t(t), _t(clip_value_min)), _t(clip_value_max))


def reduce_min(input_tensor, axis=None, keepdims=False, name=None):
    x = _t(input_tensor)
    return x.min() if axis is None else x.amin(dim=axis, keepdim=keepdims)


def bucketize(input, boundaries, name=None):
    """tf.raw_ops.Bucketize: output = number of boundaries <= input (boundaries sorted ascending), int32."""
    b = torch.as_tensor(np.asarray(boundaries, dtype=np.float64))
    return _t(torch.bucketize(_t(input).detach().to(torch.float64), b, right=True))


def softmax_cross_entropy_with_logits(labels, logits, axis=-1, name=None):
    """tf.nn.softmax_cross_entropy_with_logits(labels, logits): -sum(labels * log_softmax(logits), axis); the kernel broadcasts the two
    operands against each other (xent_op.cc: "logits and labels must be broadcastable") and backpropagates into both.  mpp.py:125 passes
    (predictions, target labels) in THAT order, i.e. labels = the predicted logits [n, 2^bits], logits = the label ids [n, 1]."""
    lab, lg = torch.broadcast_tensors(_t(labels), _t(logits))
    return -(lab * torch.log_softmax(lg, dim=axis)).sum(dim=axis)


def cast(x, dtype):
    dtype = {bool: torch.bool, float: DTYPE, int: torch.int64}.get(dtype, dtype)
    if dtype in (torch.float32, torch.float16, torch.bfloat16):   # the shim computes in float64 throughout
        dtype = DTYPE
    if isinstance(x, torch.Tensor):
        return _t(x.to(dtype))
    return _t(np.asarray(x), dtype=dtype)


def split(value, num_or_size_splits, axis=0, num=None, name=None):
    if isinstance(num_or_size_splits, int):
        n = value.shape[axis]
        assert n % num_or_size_splits == 0
        return list(torch.split(value, n // num_or_size_splits, dim=axis))
    return list(torch.split(value, list(num_or_size_splits), dim=axis))


def concat(values, axis, name=None):
    return torch.cat([_t(v) for v in values], dim=axis)


def einsum(equation, *inputs, **_):
    return torch.einsum(equation.replace(" ", ""), *[_t(i) for i in inputs])


def matmul(a, b, transpose_a=False, transpose_b=False, **_):
    a, b = _t(a), _t(b)
    if transpose_a:
        a = a.transpose(-1, -2)
    if transpose_b:
        b = b.transpose(-1, -2)
    return torch.matmul(a, b)


def transpose(a, perm=None, conjugate=False, name=None):
    if perm is None:
        perm = list(range(a.dim()))[::-1]
    return a.permute(*[int(p) for p in perm])


def _reduce(fn):
    def f(input_tensor, axis=None, keepdims=False, name=None):
        x = _t(input_tensor)
        if axis is None:
            return fn(x)
        return fn(x, dim=axis, keepdim=keepdims)
    return f


reduce_mean = _reduce(torch.mean)
reduce_sum = _reduce(torch.sum)


def fill(dims, value, name=None):
    return _t(torch.full(_shape_list(dims), float(value), dtype=DTYPE))


def identity(x, name=None):
    return x


def stop_gradient(x, name=None):
    return _t(x).detach()


def argmax(input, axis=None, output_type=None, name=None):   # ties -> lowest index (tf.argmax)
    a = _t(input).detach().numpy()
    return _t(np.argmax(a, axis=axis))


def argsort(values, axis=-1, direction="ASCENDING", stable=False, name=None):
    a = _t(values).detach().numpy()
    idx = np.argsort(a, axis=axis, kind="stable")
    if direction == "DESCENDING":
        idx = np.flip(idx, axis=axis)
    return _t(np.ascontiguousarray(idx))


def range_(start, limit=None, delta=1, dtype=None, name=None):
    if limit is None:
        start, limit = 0, start
    return _t(np.arange(start, limit, delta))


def where(condition, x=None, y=None, name=None):
    return torch.where(_t(condition, torch.bool), _t(x), _t(y))


def square(x, name=None):
    """tf.square(x, name=None): the second positional parameter is `name` (mae.py:90 passes a tensor there)."""
    return _t(x) * _t(x)


def softmax(logits, axis=-1, name=None):
    return torch.softmax(_t(logits), dim=axis)


def log_softmax(logits, axis=-1, name=None):
    return torch.log_softmax(_t(logits), dim=axis)


class _TopK:
    def __init__(self, values, indices):
        self.values, self.indices = values, indices

    def __iter__(self):          # `_, sampled_indices = tf.math.top_k(...)` (mpp.py:83)
        return iter((self.values, self.indices))


def top_k(input, k=1, sorted=True, name=None):
    v, i = torch.topk(_t(input), k, dim=-1, largest=True, sorted=True)
    return _TopK(v, _t(i))


def extract_patches(images, sizes, strides, rates, padding, name=None):
    """tf.image.extract_patches, rates 1: out[b, oy, ox, (ky, kx, c)] = in[b, oy*s - pad_top + ky, ox*s - pad_left + kx, c]
    (zero outside); SAME: out = ceil(in / s), pad_total = max((out - 1) s + k - in, 0), pad_before = pad_total // 2."""
    assert list(rates) == [1, 1, 1, 1]
    x = _t(images)
    b, H, W, C = x.shape
    kh, kw, sh, sw = int(sizes[1]), int(sizes[2]), int(strides[1]), int(strides[2])
    if padding == "SAME":
        oh, ow = -(-H // sh), -(-W // sw)
        ph, pw = max((oh - 1) * sh + kh - H, 0), max((ow - 1) * sw + kw - W, 0)
        pt, pl = ph // 2, pw // 2
    else:
        oh, ow = (H - kh) // sh + 1, (W - kw) // sw + 1
        ph = pw = pt = pl = 0
    xp = torch.zeros((b, H + ph, W + pw, C), dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W] = x
    rows = []
    for ky in range(kh):
        for kx in range(kw):
            rows.append(xp[:, ky:ky + (oh - 1) * sh + 1:sh, kx:kx + (ow - 1) * sw + 1:sw, :])
    return torch.stack(rows, dim=3).reshape(b, oh, ow, kh * kw * C)


# --------------------------------------------------------------------------------------- tf.keras.layers
def _accepts(fn, name: str) -> bool:
    try:
        sig = inspect.signature(fn)
    except (TypeError, ValueError):
        return False
    for p in sig.parameters.values():
        if p.name == name or p.kind is inspect.Parameter.VAR_KEYWORD:
            return True
    return False


class Layer:
    def __init__(self, name=None, trainable=True, dtype=None, **kwargs):
        self.name = name or type(self).__name__.lower()
        self.built = False
        self.trainable = trainable

    def build(self, input_shape):
        pass

    def add_weight(self, name=None, shape=None, initializer=None, trainable=True, dtype=None, **_):
        return Variable(initializer(shape) if callable(initializer) else initializer, trainable=trainable)

    def call(self, inputs, *args, **kwargs):
        return inputs

    def __call__(self, *args, **kwargs):
        if not self.built:
            if args and getattr(args[0], "shape", None) is not None:     # tensors and NumPy arrays alike (simmim.py:122 hands one over)
                self.build(tuple(args[0].shape))
            self.built = True
        if "training" in kwargs and not _accepts(self.call, "training"):
            kwargs.pop("training")   # Keras: a `training` argument call() does not understand is not forwarded
        return self.call(*args, **kwargs)

    @property
    def weights(self):
        return list(getattr(self, "_weights", []))


class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, **kw):
        super().__init__(**kw)
        assert activation is None
        self.units, self.use_bias = int(units), use_bias
        self.kernel = self.bias = None

    def build(self, input_shape):
        fan_in = int(input_shape[-1])
        lim = math.sqrt(6.0 / (fan_in + self.units))           # glorot_uniform
        self.kernel = Variable(_random_uniform([fan_in, self.units], -lim, lim))
        self._weights = [self.kernel]
        if self.use_bias:
            self.bias = Variable(torch.zeros(self.units, dtype=DTYPE))
            self._weights.append(self.bias)

    def call(self, inputs):
        y = torch.matmul(_t(inputs), self.kernel)                # tensordot over the last axis for rank > 2
        return y + self.bias if self.use_bias else y


class LayerNormalization(Layer):
    def __init__(self, axis=-1, epsilon=1e-3, center=True, scale=True, **kw):
        super().__init__(**kw)
        assert axis in (-1,) and center and scale
        self.epsilon = epsilon
        self.gamma = self.beta = None

    def build(self, input_shape):
        d = int(input_shape[-1])
        self.gamma = Variable(torch.ones(d, dtype=DTYPE))
        self.beta = Variable(torch.zeros(d, dtype=DTYPE))
        self._weights = [self.gamma, self.beta]

    def call(self, inputs):
        x = _t(inputs)
        mean = x.mean(dim=-1, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)      # tf.nn.moments: biased
        return (x - mean) * torch.rsqrt(var + self.epsilon) * self.gamma + self.beta


class Softmax(Layer):
    def __init__(self, axis=-1, **kw):
        super().__init__(**kw)
        self.axis = axis

    def call(self, inputs):
        return torch.softmax(_t(inputs), dim=self.axis)


class Activation(Layer):
    def __init__(self, activation, **kw):
        super().__init__(**kw)
        self.activation = activation

    def call(self, inputs):
        return self.activation(inputs)


class Dropout(Layer):
    def __init__(self, rate, **kw):
        super().__init__(**kw)
        self.rate = float(rate)

    def call(self, inputs, training=None):
        if not training or self.rate == 0.0:
            return inputs
        keep = (torch.rand(inputs.shape, generator=_GEN, dtype=DTYPE) >= self.rate).to(DTYPE)
        return inputs * keep / (1.0 - self.rate)


class Embedding(Layer):
    def __init__(self, input_dim, output_dim, **kw):
        super().__init__(**kw)
        self.input_dim, self.output_dim = int(input_dim), int(output_dim)
        self.embeddings = None

    def build(self, input_shape):
        self.embeddings = Variable(_random_uniform([self.input_dim, self.output_dim], -0.05, 0.05))   # Keras 'uniform'
        self._weights = [self.embeddings]

    def call(self, inputs):
        return self.embeddings[_t(inputs, torch.int64)]


class Sequential(Layer):
    def __init__(self, layers=None, name=None):
        super().__init__(name=name)
        self.layers = list(layers) if layers else []

    def add(self, layer):
        self.layers.append(layer)

    def call(self, inputs, training=None, mask=None):
        x = inputs
        for layer in self.layers:
            kw = {"training": training} if _accepts(layer.call, "training") else {}
            x = layer(x, **kw)
        return x


class Model(Layer):
    def build(self, input_shape):
        """Model.build(input_shape): Keras runs call() on a placeholder of that shape (mae.py:32, simmim.py:75)."""
        if getattr(self, "_model_built", False):
            return
        self._model_built = True
        self(torch.zeros(_shape_list(input_shape), dtype=DTYPE).as_subclass(_T))

    def __call__(self, *args, **kwargs):
        self.built = True
        self._model_built = True
        return self.call(*args, **kwargs)


# ------------------------------------------------------------------------------------- tf.keras.losses
def categorical_crossentropy(y_true, y_pred, from_logits=False, label_smoothing=0.0, axis=-1):
    y_true, y_pred = _t(y_true), _t(y_pred)
    if from_logits:
        return -(y_true * torch.log_softmax(y_pred, dim=axis)).sum(dim=axis)
    p = y_pred / y_pred.sum(dim=axis, keepdim=True)
    p = p.clamp(1e-7, 1 - 1e-7)
    return -(y_true * torch.log(p)).sum(dim=axis)


class _Reduction:
    NONE = "none"
    SUM = "sum"
    AUTO = "auto"
    SUM_OVER_BATCH_SIZE = "sum_over_batch_size"


class KLDivergence:
    """keras.losses.KLDivergence: y_true, y_pred clipped to [1e-7, 1]; loss = sum(y_true * log(y_true / y_pred), axis=-1)."""

    def __init__(self, reduction="auto", name="kl_divergence"):
        self.reduction = reduction

    def __call__(self, y_true, y_pred, sample_weight=None):
        yt = _t(y_true).clamp(1e-7, 1.0)
        yp = _t(y_pred).clamp(1e-7, 1.0)
        loss = (yt * torch.log(yt / yp)).sum(dim=-1)
        if self.reduction == _Reduction.NONE:
            return loss
        if self.reduction == _Reduction.SUM:
            return loss.sum()
        return loss.mean()


# ------------------------------------------------------------------------------------------ install()
def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []          # importable as a package (`import tensorflow.keras.layers as nn`)
    return m


def install() -> types.ModuleType:
    """Register the stand-in under sys.modules['tensorflow'...] (idempotent) and return the top module."""
    if "tensorflow" in sys.modules and getattr(sys.modules["tensorflow"], "__vitx_shim__", False):
        return sys.modules["tensorflow"]
    assert "tensorflow" not in sys.modules, "a real TensorFlow is importable here: use it instead of the shim"
    layers = _module("tensorflow.keras.layers", Layer=Layer, Dense=Dense, LayerNormalization=LayerNormalization,
                     Softmax=Softmax, Activation=Activation, Dropout=Dropout, Embedding=Embedding)
    losses = _module("tensorflow.keras.losses", categorical_crossentropy=categorical_crossentropy,
                     KLDivergence=KLDivergence, Reduction=_Reduction)
    backend = _module("tensorflow.keras.backend", is_keras_tensor=lambda x: False)   # einops probes it while choosing a backend
    keras = _module("tensorflow.keras", Model=Model, Sequential=Sequential, layers=layers, losses=losses, backend=backend)
    nn_ = _module("tensorflow.nn", softmax=softmax, log_softmax=log_softmax, softmax_cross_entropy_with_logits=softmax_cross_entropy_with_logits)
    raw_ops = _module("tensorflow.compat.v1.raw_ops", Bucketize=bucketize)
    compat_v1 = _module("tensorflow.compat.v1", raw_ops=raw_ops)
    compat = _module("tensorflow.compat", v1=compat_v1)
    math_ = _module("tensorflow.math", erf=lambda x: torch.erf(_t(x)), top_k=top_k, tanh=lambda x: torch.tanh(_t(x)))
    random_ = _module("tensorflow.random", normal=_random_normal, uniform=_random_uniform)
    image_ = _module("tensorflow.image", extract_patches=extract_patches)
    tf = _module("tensorflow", __vitx_shim__=True, __version__="0.0-vitx-shim", Tensor=Tensor, Variable=Variable,
                 keras=keras, nn=nn_, math=math_, random=random_, image=image_, compat=compat, zeros=zeros, expand_dims=expand_dims, reshape=reshape,
                 clip_by_value=clip_by_value, reduce_min=reduce_min,
                 float32=torch.float32, float64=torch.float64, int32=torch.int32, int64=torch.int64, bool=torch.bool,
                 cast=cast, split=split, concat=concat, einsum=einsum, matmul=matmul, transpose=transpose,
                 reduce_mean=reduce_mean, reduce_sum=reduce_sum, fill=fill, identity=identity, stop_gradient=stop_gradient,
                 argmax=argmax, argsort=argsort, range=range_, where=where, square=square, abs=lambda x: torch.abs(_t(x)),
                 tanh=lambda x: torch.tanh(_t(x)), pow=lambda x, y: torch.pow(_t(x), y), sqrt=lambda x: torch.sqrt(_t(x)),
                 convert_to_tensor=_t, constant=_t, executing_eagerly=lambda: True, is_tensor=lambda x: False)
    for m in (tf, keras, layers, losses, backend, nn_, math_, random_, image_, compat, compat_v1, raw_ops):
        sys.modules[m.__name__] = m
    return tf


def uninstall() -> None:
    for k in [k for k in sys.modules if k == "tensorflow" or k.startswith("tensorflow.")]:
        if getattr(sys.modules.get("tensorflow"), "__vitx_shim__", False) or k != "tensorflow":
            sys.modules.pop(k, None)
    sys.modules.pop("einops.layers.tensorflow", None)
