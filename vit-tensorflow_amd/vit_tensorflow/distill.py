"""Drop-in for vit_tensorflow/distill.py: `DistillableViT` (a ViT whose call accepts a distillation token, distill.py:16-57) and
`DistillWrapper(teacher, student, temperature, alpha, hard)` (distill.py:87-134) on the MI355X engine.  The teacher is any
callable returning logits (another engine model, a numpy function, ...): it runs outside, its logits are an input."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import numpy as np

from . import _native as N
from ._model import VitxModel, _Weight
from .t2t import T2TViT
from .vit import ViT


def exists(val):
    """distill.py:12-13"""
    return val is not None


class DistillableViT(ViT):
    """distill.py:46-57: same constructor as ViT; `model(img, distill_token=tok)` returns (logits, distill_tokens)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.args = args
        self.kwargs = kwargs
        self.dim = kwargs['dim'] if 'dim' in kwargs else self.dim
        self.num_classes = kwargs['num_classes'] if 'num_classes' in kwargs else self.num_classes

    def __call__(self, img, distill_token=None, training=True, **kw):
        if not exists(distill_token):
            return super().__call__(img, training=training, **kw)
        x, proto = self._as_host(img)
        assert x.ndim == 4, "expected NHWC images [b, H, W, C]"
        b, H, W, Cc = x.shape
        assert H % self._cfg.patch_h == 0 and W % self._cfg.patch_w == 0, 'Image dimensions must be divisible by the patch size.'
        tok = np.ascontiguousarray(np.asarray(distill_token, dtype=np.float32).reshape(-1))
        assert tok.size == self.dim, "distill_token must have shape [1, 1, dim]"
        h = self._ensure_handle(b)
        self._img_shape = (b, H, W, Cc)
        logits = np.empty((b, self.num_classes), dtype=np.float32)
        dtok = np.empty((b, self.dim), dtype=np.float32)
        seed = int(kw.get("seed", np.random.randint(0, 2 ** 31 - 1)))
        N.check(N.lib().vitx_forward_distill(h, x.ctypes.data_as(C.c_void_p), b, H, W, 1 if training else 0, seed,
                                             tok.ctypes.data_as(C.c_void_p), logits.ctypes.data_as(C.c_void_p), dtok.ctypes.data_as(C.c_void_p)))
        return self._like(logits, proto), self._like(dtok, proto)

    call = __call__

    def backward_distill(self, dlogits, d_distill_tokens=None):
        """VJP of the last `model(img, distill_token=tok)`: ({name: grad}, d(distill_token) [1, 1, dim])."""
        if self._handle is None:
            raise N.VitxError(N.ERR_STATE, "backward requires a preceding forward")
        d, _ = self._as_host(dlogits)
        dd = None if d_distill_tokens is None else self._as_host(d_distill_tokens)[0]
        dt = np.empty(self.dim, dtype=np.float32)
        N.check(N.lib().vitx_backward_distill(self._handle, d.ctypes.data_as(C.c_void_p), None if dd is None else dd.ctypes.data_as(C.c_void_p),
                                              dt.ctypes.data_as(C.c_void_p), None))
        self._finish_exchange()
        g = np.empty(self._n, dtype=np.float32)
        N.check(N.lib().vitx_get_grads(self._handle, g.ctypes.data_as(C.c_void_p), self._n))
        return {n: g[o:o + int(np.prod(s))].reshape(s) for n, s, o in self._table}, dt.reshape(1, 1, -1)


class DistillableT2TViT(T2TViT):
    """distill.py:60-72: a T2TViT whose call accepts a distillation token.  The tokenizer runs as in T2TViT; the token joins the
    sequence on the main engine handle, which reads the tokenizer's output as patch rows (vitx_set_patch_input)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.args = args
        self.kwargs = kwargs

    # what DistillWrapper needs from a student: the handle that carries the distillation token is the main one
    @property
    def _handle(self):
        return self._main._handle

    @property
    def _cfg(self):
        return self._main._cfg

    @property
    def _handle_gen(self):
        return self._main._handle_gen

    def _ensure_handle(self, b):
        return self._main._ensure_handle(b)

    def __call__(self, img, distill_token=None, training=True, **kw):
        if not exists(distill_token):
            return super().__call__(img, training=training, **kw)
        x, proto = VitxModel._as_host(img)
        tok_in = self._tokenize(x, training, kw.get("seed"))                       # [b, n, last layer_dim]
        b, n, _ = tok_in.shape
        tok = np.ascontiguousarray(np.asarray(distill_token, dtype=np.float32).reshape(-1))
        assert tok.size == self.dim, "distill_token must have shape [1, 1, dim]"
        m = self._main
        h = m._ensure_handle(b)
        m._img_shape = (b, n, tok_in.shape[2])
        logits = np.empty((b, self.num_classes), dtype=np.float32)
        dtok = np.empty((b, self.dim), dtype=np.float32)
        a = np.ascontiguousarray(tok_in, dtype=np.float32)
        N.check(N.lib().vitx_set_patch_input(h, n))
        N.check(N.lib().vitx_forward_distill(h, a.ctypes.data_as(C.c_void_p), b, 0, 0, 1 if training else 0, self._last_seed,
                                             tok.ctypes.data_as(C.c_void_p), logits.ctypes.data_as(C.c_void_p), dtok.ctypes.data_as(C.c_void_p)))
        return VitxModel._like(logits, proto), VitxModel._like(dtok, proto)

    call = __call__

    def backward_distill(self, dlogits, d_distill_tokens=None):
        """VJP of the last `model(img, distill_token=tok)`: ({name: grad}, d(distill_token) [1, 1, dim])."""
        m = self._main
        d, _ = VitxModel._as_host(dlogits)
        dd = None if d_distill_tokens is None else VitxModel._as_host(d_distill_tokens)[0]
        dt = np.empty(self.dim, dtype=np.float32)
        dpat = np.empty(m._img_shape, dtype=np.float32)
        N.check(N.lib().vitx_backward_distill(m._handle, d.ctypes.data_as(C.c_void_p), None if dd is None else dd.ctypes.data_as(C.c_void_p),
                                              dt.ctypes.data_as(C.c_void_p), dpat.ctypes.data_as(C.c_void_p)))
        m._finish_exchange()
        g = np.empty(m._n, dtype=np.float32)
        N.check(N.lib().vitx_get_grads(m._handle, g.ctypes.data_as(C.c_void_p), m._n))
        gm = {n: g[o:o + int(np.prod(s))].reshape(s) for n, s, o in m._table}
        return self._chain_tokenizer(gm, dpat, False)[0], dt.reshape(1, 1, -1)


class DistillableEfficientViT:
    """distill.py:74-85.  The reference's class cannot run: DistillMixin.call reaches `self.dropout` through `_attend`
    (distill.py:83), an attribute efficient.ViT never defines (efficient.py:13-36), so every call raises AttributeError there.
    The drop-in keeps that behaviour: it constructs (the shell is vit_tensorflow.efficient.ViT) and fails on call with the
    reference's own error."""

    def __init__(self, *args, **kwargs):
        from .efficient import ViT as EfficientViT
        self._shell = EfficientViT(*args, **kwargs)
        self.args, self.kwargs = args, kwargs
        self.dim, self.num_classes = kwargs['dim'], kwargs['num_classes']

    def __call__(self, img, distill_token=None, training=True):
        raise AttributeError("'DistillableEfficientViT' object has no attribute 'dropout'")

    call = __call__


class DistillWrapper:
    def __init__(self, teacher, student, temperature=1.0, alpha=0.5, hard=False, *, literal_loss=True, seed=None):
        """Same arguments as the reference (distill.py:88).  Engine-only keyword extras: literal_loss (soft mode) -- True keeps
        the distillation term exactly as distill.py:122-129 computes it (Keras' KLDivergence clips the LOG-probabilities it is
        handed to 1e-7, which makes the term constant in the student), False computes the intended KL divergence; seed."""
        assert isinstance(student, (DistillableViT, DistillableT2TViT, DistillableEfficientViT)), 'student must be a vision transformer'   # distill.py:91
        self._t2t = isinstance(student, DistillableT2TViT)
        self.teacher, self.student = teacher, student
        self.temperature, self.alpha, self.hard = temperature, alpha, hard
        # a DistillableEfficientViT student: the reference CONSTRUCTS the wrapper and fails on its first call, inside the student's call, with
        # AttributeError('dropout') (distill.py:83 / the class above) -- same here: nothing engine-side is built, __call__ invokes the student
        self._unrunnable = isinstance(student, DistillableEfficientViT)
        if self._unrunnable:
            self._h = None
            return
        cfg = N.DistillConfig()
        cfg.temperature, cfg.alpha, cfg.hard, cfg.literal_loss = float(temperature), float(alpha), 1 if hard else 0, 1 if literal_loss else 0
        self._dcfg = cfg
        self._h: Optional[C.c_void_p] = None
        self._stu_gen = -1
        self._rng = np.random.default_rng(seed)
        self._table: List = []
        self._n = 0
        self._blob: Optional[np.ndarray] = None
        self._device_newer = False
        self._ensure(1)

    def _ensure(self, batch: int):
        l = N.lib()
        stu = self.student
        rebuild = stu._handle is None or batch > stu._cfg.max_batch
        stale = self._h is not None and self._stu_gen != stu._handle_gen
        if self._h is not None and (rebuild or stale):
            N.check(l.vitx_distill_destroy(self._h))     # before the student handle it points into goes away
            self._h = None
        sh = stu._ensure_handle(batch)
        if self._h is not None:
            return self._h
        h = C.c_void_p()
        N.check(l.vitx_distill_create(sh, C.byref(self._dcfg), C.byref(h)))
        self._h, self._stu_gen = h, stu._handle_gen
        self._table, self._n = N.mim_param_table(h, "vitx_distill")
        if self._blob is None:
            self._blob = np.zeros(self._n, dtype=np.float32)
            self._init_weights()
        self._push_params()
        return h

    # tf.random.normal token (distill.py:101); LayerNormalization ones / zeros; Dense glorot_uniform / zeros (distill.py:103-106)
    def _init_weights(self):
        for name, shape, off in self._table:
            n = int(np.prod(shape))
            leaf = name.split(".")[-1]
            if name == "distillation_token":
                v = self._rng.standard_normal(n)
            elif leaf == "kernel":
                lim = math.sqrt(6.0 / (shape[0] + shape[1]))
                v = self._rng.uniform(-lim, lim, n)
            elif leaf == "gamma":
                v = np.ones(n)
            else:
                v = np.zeros(n)
            self._blob[off:off + n] = v.astype(np.float32)

    def _push_params(self):
        if self._h is not None:
            N.check(N.lib().vitx_distill_set_params(self._h, self._blob.ctypes.data_as(C.c_void_p), self._n))
        self._device_newer = False

    def _pull_params(self):
        pass   # the wrapper's variables only change through set_weights / assign (host side)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                N.lib().vitx_distill_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def weights(self) -> List[_Weight]:
        return [_Weight(self, n, s, o) for n, s, o in self._table]

    @property
    def distillation_token(self):
        return self.weights[0]

    def get_weights(self) -> List[np.ndarray]:
        return [self._blob[o:o + int(np.prod(s))].reshape(s).copy() for _, s, o in self._table]

    def set_weights(self, weights) -> None:
        assert len(weights) == len(self._table), f"expected {len(self._table)} arrays, got {len(weights)}"
        for w, (n, s, o) in zip(weights, self._table):
            a = np.asarray(w, dtype=np.float32)
            assert a.shape == tuple(s), f"{n}: expected shape {tuple(s)}, got {a.shape}"
            self._blob[o:o + a.size] = a.reshape(-1)
        self._push_params()

    def state_dict(self) -> Dict[str, np.ndarray]:
        return {n: w for (n, _, _), w in zip(self._table, self.get_weights())}

    def load_state_dict(self, sd) -> None:
        self.set_weights([sd[n] for n, _, _ in self._table])

    def __call__(self, inputs, temperature=None, alpha=None, training=True, **kwargs):
        """DistillWrapper.call((img, labels), temperature, alpha, training) (distill.py:107): the per-image loss [b]."""
        img, labels = inputs
        if self._unrunnable:   # distill.py:116: student(img, distill_token=..., training=...) -> AttributeError('dropout') inside the student's call
            return self.student(img, distill_token=None, training=training)
        x, _ = VitxModel._as_host(img)
        y, _ = VitxModel._as_host(labels)
        b, H, W, _c = x.shape
        seed = int(kwargs.get("seed", np.random.randint(0, 2 ** 31 - 1)))
        assert y.shape == (b, self.student.num_classes), "labels must be [b, num_classes] (one-hot or soft)"
        t, _ = VitxModel._as_host(self.teacher(img, training=training) if not isinstance(self.teacher, np.ndarray) else self.teacher)   # distill.py:114
        assert t.shape == y.shape, "teacher must return logits [b, num_classes]"
        h = self._ensure(b)
        loss = np.empty(b, dtype=np.float32)
        if self._t2t:   # the student's tokenizer runs in front of its main handle, which then reads patch rows
            x = np.ascontiguousarray(self.student._tokenize(x, training, seed), dtype=np.float32)
            self.student._main._img_shape = x.shape
            N.check(N.lib().vitx_set_patch_input(self.student._handle, x.shape[1]))
            H = W = 0
        N.check(N.lib().vitx_distill_forward(h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), b, H, W,
                                             1 if training else 0, seed, -1.0 if temperature is None else float(temperature),
                                             -1.0 if alpha is None else float(alpha), loss.ctypes.data_as(C.c_void_p)))
        return loss

    call = __call__

    def backward(self, dloss=None) -> Dict[str, np.ndarray]:
        """Gradients of the last loss for the cotangent dloss [b] (default ones: tape.gradient of a vector target sums it):
        wrapper variables under their own names, the student's under 'student.<name>'."""
        if self._h is None or self._stu_gen != self.student._handle_gen:
            raise N.VitxError(N.ERR_STATE, "backward requires a preceding forward")
        l = N.lib()
        stu = self.student._main if self._t2t else self.student
        if hasattr(stu, "_refuse_exchange"):
            stu._refuse_exchange("DistillWrapper.backward")
        dl = None if dloss is None else np.ascontiguousarray(np.asarray(dloss, dtype=np.float32))
        dpat = None
        if self._t2t:
            dpat = np.empty(self.student._main._img_shape, dtype=np.float32)
            N.check(l.vitx_distill_backward_input(self._h, None if dl is None else dl.ctypes.data_as(C.c_void_p), dpat.ctypes.data_as(C.c_void_p)))
        else:
            N.check(l.vitx_distill_backward(self._h, None if dl is None else dl.ctypes.data_as(C.c_void_p)))
        g = np.empty(self._n, dtype=np.float32)
        N.check(l.vitx_distill_get_grads(self._h, g.ctypes.data_as(C.c_void_p), self._n))
        out = {n: g[o:o + int(np.prod(s))].reshape(s) for n, s, o in self._table}
        if self._t2t:
            m = self.student._main
            sg = np.empty(m._n, dtype=np.float32)
            N.check(l.vitx_get_grads(m._handle, sg.ctypes.data_as(C.c_void_p), m._n))
            gm = {n: sg[o:o + int(np.prod(s))].reshape(s) for n, s, o in m._table}
            for n, a in self.student._chain_tokenizer(gm, dpat, False)[0].items():
                out["student." + n] = a
            return out
        stu = self.student
        sg = np.empty(stu._n, dtype=np.float32)
        N.check(l.vitx_get_grads(stu._handle, sg.ctypes.data_as(C.c_void_p), stu._n))
        for n, s, o in stu._table:
            out["student." + n] = sg[o:o + int(np.prod(s))].reshape(s)
        return out

    def read(self, which: str) -> np.ndarray:
        n = C.c_int64()
        l = N.lib()
        l.vitx_distill_read(self._h, which.encode(), np.empty(1, np.float32).ctypes.data_as(C.c_void_p), 0, C.byref(n))
        out = np.empty(max(int(n.value), 1), dtype=np.float32)
        N.check(l.vitx_distill_read(self._h, which.encode(), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out[:n.value]
