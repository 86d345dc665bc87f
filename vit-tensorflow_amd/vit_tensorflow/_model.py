"""Host-side mirror of the reference's model objects (tf.keras.Model subclasses in
vit_tensorflow/vit.py:106, deepvit.py:112, cait.py:155): same constructor kwargs, same call
signature, same assertion texts.  All arithmetic is delegated to libvitx (HIP) through ctypes."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _native as N


def pair(t):
    """vit.py:11-12"""
    return t if isinstance(t, tuple) else (t, t)


def layer_scale_init(depth_1based: int) -> float:
    """cait.py:36-41"""
    if depth_1based <= 18:
        return 0.1
    if depth_1based <= 24:
        return 1e-5
    return 1e-6


class _Weight:
    """Minimal stand-in for a tf.Variable: `.name`, `.shape`, `.numpy()`, `.assign()`."""

    def __init__(self, owner: "VitxModel", name: str, shape, offset: int):
        self._owner, self.name, self.shape, self._offset = owner, name, tuple(shape), offset

    def numpy(self) -> np.ndarray:
        self._owner._pull_params()
        n = int(np.prod(self.shape))
        return self._owner._blob[self._offset:self._offset + n].reshape(self.shape).copy()

    def assign(self, value) -> None:
        v = np.asarray(value, dtype=np.float32)
        assert v.shape == self.shape, f"shape mismatch for {self.name}: {v.shape} vs {self.shape}"
        self._owner._pull_params()
        self._owner._blob[self._offset:self._offset + v.size] = v.reshape(-1)
        self._owner._push_params()

    def __array__(self, dtype=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, idx):
        """`encoder.pos_embedding[:, 1:(num_patches + 1)]` (mae.py:54, simmim.py:95, distill.py:24)."""
        return self.numpy()[idx]


class _TransformerProxy:
    """`model.transformer(tokens, training=...)` as used by mae.py:69, simmim.py:116, mpp.py:212."""

    def __init__(self, owner: "VitxModel"):
        self._owner = owner

    def __call__(self, tokens, training=True, seed=None, **_):
        return self._owner._transformer_call(tokens, training, seed)

    def backward(self, dout, want_dtokens: bool = True):
        """VJP of the last `transformer(tokens)` call: returns ({name: grad}, dtokens|None); gradients of parameters outside the
        transformer are zero."""
        return self._owner._transformer_backward(dout, want_dtokens)


class _ToPatch:
    """`encoder.patch_embedding.layers[0]`: Rearrange('b (h p1) (w p2) c -> b (h w) (p1 p2 c)') (vit.py:142; borrowed at mae.py:37,
    simmim.py:79).  Bit-exact HIP gather."""

    def __init__(self, owner: "VitxModel"):
        self._owner = owner

    def __call__(self, img, training=None, **_):
        o = self._owner
        x, proto = o._as_host(img)
        assert x.ndim == 4, "expected NHWC images [b, H, W, C]"
        b, H, W, Cc = x.shape
        ph, pw = o._cfg.patch_h, o._cfg.patch_w
        assert H % ph == 0 and W % pw == 0, 'Image dimensions must be divisible by the patch size.'
        out = np.empty((b, (H // ph) * (W // pw), ph * pw * Cc), dtype=np.float32)
        N.check(N.lib().vitx_patch_unfold(x.ctypes.data_as(C.c_void_p), b, H, W, Cc, ph, pw, out.ctypes.data_as(C.c_void_p)))
        return o._like(out, proto)


class _PatchToEmb:
    """`encoder.patch_embedding.layers[1]`: nn.Dense(units=dim) (vit.py:143); `.weights[0].shape` is read at mae.py:38 / simmim.py:80."""

    def __init__(self, owner: "VitxModel"):
        self._owner = owner

    @property
    def weights(self):
        return [w for w in self._owner.weights if w.name in ("patch_embedding.kernel", "patch_embedding.bias")]

    def __call__(self, patches, training=None, **_):
        o = self._owner
        x, proto = o._as_host(patches)
        pd = o._cfg.patch_h * o._cfg.patch_w * o._cfg.channels
        assert x.shape[-1] == pd, f"expected patches of {pd} features"
        rows = int(np.prod(x.shape[:-1]))
        npatch = (o._cfg.image_h // o._cfg.patch_h) * (o._cfg.image_w // o._cfg.patch_w)
        h = o._ensure_handle(max(1, -(-rows // npatch)))
        out = np.empty(x.shape[:-1] + (o.dim,), dtype=np.float32)
        N.check(N.lib().vitx_patch_dense_forward(h, x.ctypes.data_as(C.c_void_p), rows, out.ctypes.data_as(C.c_void_p)))
        return o._like(out, proto)


class _PatchEmbedding:
    """`model.patch_embedding` (a Keras Sequential in the reference, vit.py:141-144): callable, with `.layers` (distill.py:19,
    mae.py:37, mpp.py:200)."""

    def __init__(self, owner: "VitxModel"):
        self.layers = [_ToPatch(owner), _PatchToEmb(owner)]

    def __call__(self, img, training=None, **_):
        return self.layers[1](self.layers[0](img))


class _MlpHead:
    """`model.mlp_head` (vit.py:154-157): LayerNormalization + Dense on pooled features [b, dim] (distill.py:40)."""

    def __init__(self, owner: "VitxModel"):
        self._owner = owner

    def __call__(self, x, training=None, **_):
        o = self._owner
        a, proto = o._as_host(x)
        assert a.ndim == 2 and a.shape[1] == o.dim, "expected pooled features [b, dim]"
        b = a.shape[0]
        h = o._ensure_handle(b)
        out = np.empty((b, o.num_classes), dtype=np.float32)
        # one token per image: cls and mean pooling are both the identity on it, so this is exactly LayerNorm + Dense
        N.check(N.lib().vitx_head_forward(h, a.ctypes.data_as(C.c_void_p), b, 1, out.ctypes.data_as(C.c_void_p)))
        return o._like(out, proto)


class _Dropout:
    """`model.dropout` (nn.Dropout(rate=emb_dropout), vit.py:148) as called by distill.py:56 / mpp.py:209.  Inside the engine's own
    forward the mask comes from the device-side counter RNG; this stand-alone layer is the identity unless a rate > 0 is asked to
    act in training mode, which only the full forward offers."""

    def __init__(self, rate: float):
        self.rate = float(rate)

    def __call__(self, x, training=False, **_):
        if training and self.rate > 0.0:
            raise NotImplementedError("stand-alone dropout with rate > 0: use model(img, training=True) (the mask is drawn inside the engine)")
        return x


class VitxModel:
    _variant = N.VARIANT_VIT

    def _init_common(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool, dim_head, dropout,
                     emb_dropout, layer_dropout=0.0, cls_depth=0, num_parallel_branches=1, patch_merge_layer=None, patch_merge_num_tokens=8,
                     compute="fp32", max_batch=None, device=0, seed=None, channels=3):
        ih, iw = pair(image_size)
        ph, pw = pair(patch_size)
        assert ih % ph == 0 and iw % pw == 0, 'Image dimensions must be divisible by the patch size.'
        if self._variant not in (N.VARIANT_CAIT, N.VARIANT_PATCH_MERGER):
            assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
        self.pool = pool
        self.heads, self.dim, self.depth, self.mlp_dim, self.dim_head = heads, dim, depth, mlp_dim, dim_head
        self.num_classes = num_classes
        cfg = N.Config()
        cfg.variant = self._variant
        cfg.image_h, cfg.image_w, cfg.patch_h, cfg.patch_w, cfg.channels = ih, iw, ph, pw, int(channels)
        cfg.num_classes, cfg.dim, cfg.depth, cfg.cls_depth = num_classes, dim, depth, cls_depth
        cfg.heads, cfg.dim_head, cfg.mlp_dim = heads, dim_head, mlp_dim
        cfg.pool = N.POOL_MEAN if pool == 'mean' else N.POOL_CLS
        cfg.dropout, cfg.emb_dropout, cfg.layer_dropout = float(dropout), float(emb_dropout), float(layer_dropout)
        cfg.ln_eps = 1e-3  # Keras LayerNormalization default
        assert compute in ("fp32", "bf16", "bf16x3"), "compute must be 'fp32' (parity), 'bf16' (throughput) or 'bf16x3' (fp32 data path, split-operand bf16 GEMMs)"
        cfg.compute = {"fp32": N.COMPUTE_FP32, "bf16": N.COMPUTE_BF16, "bf16x3": N.COMPUTE_BF16X3}[compute]
        cfg.num_parallel_branches = int(num_parallel_branches)
        cfg.patch_merge_layer = int(patch_merge_layer or 0)
        cfg.patch_merge_num_tokens = int(patch_merge_num_tokens)
        cfg.max_batch = int(max_batch or 0)
        cfg.device_id = int(device)
        self._cfg = cfg
        self.compute = compute
        self._handle: Optional[C.c_void_p] = None
        self._handle_gen = 0                           # bumped whenever the device plan is (re)built: wrappers hanging off it follow
        self._borrowed = False                         # True: the handle belongs to a wrapper object (MAE's decoder)
        self._table, self._n = N.param_table(cfg)      # C library is the single source of the order
        self._blob = np.zeros(self._n, dtype=np.float32)
        self._device_newer = False
        self._init_weights(np.random.default_rng(seed))
        self.transformer = _TransformerProxy(self)
        # the rest of the surface the reference's wrappers reach into: distill.py:19-40, mae.py:36-38, mpp.py:200-209
        self.patch_embedding = _PatchEmbedding(self)
        self.mlp_head = _MlpHead(self)
        self.dropout = _Dropout(emb_dropout)
        self._cb_keepalive = None

    # ---- initialisers: tf.random.normal (vit.py:146-147), Keras Dense glorot_uniform / zeros,
    #      LayerNormalization ones / zeros, LayerScale tf.fill (cait.py:43)
    def _init_weights(self, rng: np.random.Generator) -> None:
        for name, shape, off in self._table:
            n = int(np.prod(shape))
            leaf = name.split(".")[-1]
            if name in ("pos_embedding", "cls_token") or leaf in ("reattn_weights", "mix_heads_pre_attn", "mix_heads_post_attn", "queries"):
                v = rng.standard_normal(n)
            elif leaf == "kernel":
                lim = math.sqrt(6.0 / (shape[0] + shape[1]))
                v = rng.uniform(-lim, lim, n)
            elif leaf == "gamma":
                v = np.ones(n)
            elif leaf == "scale":
                ind = int(name.split(".")[1])
                v = np.full(n, layer_scale_init(ind + 1))
            else:  # bias / beta
                v = np.zeros(n)
            self._blob[off:off + n] = v.astype(np.float32)

    # ---- handle management
    def _ensure_handle(self, batch: int):
        l = N.lib()
        if self._handle is not None and batch <= self._cfg.max_batch:
            return self._handle
        if self._borrowed:
            raise N.VitxError(N.ERR_INVALID, "batch must be in [1, max_batch] (this model's device plan belongs to its wrapper)")
        opt = None
        if self._handle is not None:   # grow: keep the weights AND the optimizer state (both live in the handle), rebuild the device plan
            self._pull_params()
            opt = self._pull_opt_state()
            N.check(l.vitx_destroy(self._handle))
            self._handle = None
        self._cfg.max_batch = max(int(batch), int(self._cfg.max_batch))
        h = C.c_void_p()
        N.check(l.vitx_create(C.byref(self._cfg), C.byref(h)))
        self._handle = h
        self._handle_gen += 1
        self._push_params()
        if opt is not None:
            mom, var, step = opt
            N.check(l.vitx_set_opt_state(h, mom.ctypes.data_as(C.c_void_p) if mom is not None else None,
                                         var.ctypes.data_as(C.c_void_p) if var is not None else None, self._n, step))
        return h

    def _pull_opt_state(self):
        """(momentum | None, second moment | None, step) of the live handle; None when no optimizer step has run on it."""
        l = N.lib()
        step, hm, hv = C.c_int64(), C.c_int32(), C.c_int32()
        N.check(l.vitx_get_opt_state(self._handle, None, None, self._n, C.byref(step), C.byref(hm), C.byref(hv)))
        if not (hm.value or hv.value or step.value):
            return None
        mom = np.empty(self._n, dtype=np.float32) if hm.value else None
        var = np.empty(self._n, dtype=np.float32) if hv.value else None
        N.check(l.vitx_get_opt_state(self._handle, mom.ctypes.data_as(C.c_void_p) if mom is not None else None,
                                     var.ctypes.data_as(C.c_void_p) if var is not None else None, self._n, C.byref(step), None, None))
        return mom, var, int(step.value)

    def _push_params(self):
        if self._handle is not None:
            N.check(N.lib().vitx_set_params(self._handle, self._blob.ctypes.data_as(C.c_void_p), self._n))
        self._device_newer = False

    def _pull_params(self):
        if self._handle is not None and self._device_newer:
            N.check(N.lib().vitx_get_params(self._handle, self._blob.ctypes.data_as(C.c_void_p), self._n))
            self._device_newer = False

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None and not getattr(self, "_borrowed", False):
                N.lib().vitx_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    # ---- Keras-like surface
    def build(self, input_shape=None):
        """tf.keras.Model.build(input_shape) as called by mae.py:32: creates the device plan."""
        b = 1
        if input_shape is not None and input_shape[0]:
            b = int(input_shape[0])
        self._ensure_handle(b)
        return self

    @property
    def weights(self) -> List[_Weight]:
        return [_Weight(self, n, s, o) for n, s, o in self._table]

    trainable_variables = weights
    trainable_weights = weights

    def get_weights(self) -> List[np.ndarray]:
        self._pull_params()
        return [self._blob[o:o + int(np.prod(s))].reshape(s).copy() for _, s, o in self._table]

    def set_weights(self, weights: Sequence[np.ndarray]) -> None:
        assert len(weights) == len(self._table), f"expected {len(self._table)} arrays, got {len(weights)}"
        for w, (n, s, o) in zip(weights, self._table):
            a = np.asarray(w, dtype=np.float32)
            assert a.shape == tuple(s), f"{n}: expected shape {tuple(s)}, got {a.shape}"
            self._blob[o:o + a.size] = a.reshape(-1)
        self._push_params()

    def state_dict(self) -> Dict[str, np.ndarray]:
        return {n: w for (n, _, _), w in zip(self._table, self.get_weights())}

    def load_state_dict(self, sd: Dict[str, np.ndarray]) -> None:
        self.set_weights([sd[n] for n, _, _ in self._table])

    @staticmethod
    def _npz_path(path: str) -> str:
        path = str(path)
        return path if path.endswith(".npz") else path + ".npz"    # np.savez appends the suffix: both directions agree on the name

    def save_weights(self, path: str, format: str = "named") -> None:
        """format="named" (default): weights by engine parameter name in one .npz.  format="keras_list": the arrays of `get_weights()` in
        order as arr_0, arr_1, ... -- the form `np.savez(path, *keras_model.get_weights())` writes on the TensorFlow side.  The order is
        this library's one rule: a model's own variables first -- pos_embedding, cls_token -- then its layers in the
        attribute order of the reference constructor, kernel before bias, gamma before beta.  That is Keras 2's `Layer.trainable_weights`
        as far as it can be stated WITHOUT TensorFlow in this environment -- unverified; a list in another order fails set_weights' shape
        checks rather than loading silently, and `python -m oracle.gen_ref_fixtures --real-tf` prints Keras' actual order next to this one
        wherever TensorFlow exists.  The by-name form ("named") does not depend on any order.  NOT the TF-checkpoint / H5 files the reference's
        inherited Model.save_weights writes: those formats need TensorFlow / h5py, neither of which exists in this environment."""
        if format == "keras_list":
            np.savez(self._npz_path(path), *self.get_weights())
        elif format == "named":
            np.savez(self._npz_path(path), **self.state_dict())
        else:
            raise ValueError("format must be 'named' or 'keras_list'")

    def load_weights(self, path: str) -> None:
        """Either .npz form of save_weights (recognised by its keys)."""
        with np.load(self._npz_path(path)) as z:
            files = list(z.files)
            if files and all(f.startswith("arr_") and f[4:].isdigit() for f in files):
                self.set_weights([z[f"arr_{i}"] for i in range(len(files))])     # Keras get_weights() order
            else:
                self.load_state_dict({k: z[k] for k in files})

    @property
    def pos_embedding(self):
        return self.weights[0]

    @property
    def cls_token(self):
        return self.weights[1]

    def count_params(self) -> int:
        return int(self._n)

    # ---- tensors in / out: numpy, torch (CPU or ROCm) -- TensorFlow is what the reference used (vit.py:193)
    @staticmethod
    def _as_host(x):
        is_torch = type(x).__module__.startswith("torch")
        if is_torch:
            return np.ascontiguousarray(x.detach().to("cpu").float().numpy()), x
        return np.ascontiguousarray(np.asarray(x, dtype=np.float32)), None

    @staticmethod
    def _like(out: np.ndarray, proto):
        if proto is None:
            return out
        import torch
        return torch.from_numpy(out).to(proto.device)

    def __call__(self, img, training=True, **kwargs):
        """ViT.call(img, training=True)  (vit.py:159; `training=True` is the reference's default)."""
        x, proto = self._as_host(img)
        assert x.ndim == 4, "expected NHWC images [b, H, W, C]"
        b, H, W, Cc = x.shape
        assert Cc == self._cfg.channels, f"expected {self._cfg.channels} channels"
        assert H % self._cfg.patch_h == 0 and W % self._cfg.patch_w == 0, 'Image dimensions must be divisible by the patch size.'
        h = self._ensure_handle(b)
        self._img_shape = (b, H, W, Cc)
        out = np.empty((b, self.num_classes), dtype=np.float32)
        seed = int(kwargs.get("seed", np.random.randint(0, 2 ** 31 - 1)))
        N.check(N.lib().vitx_forward(h, x.ctypes.data_as(C.c_void_p), b, H, W, 1 if training else 0, seed,
                                     out.ctypes.data_as(C.c_void_p)))
        return self._like(out, proto)

    call = __call__
    predict = lambda self, img, **kw: self(img, training=False, **kw)

    def forward_patches(self, patches, training=True, seed=None):
        """The model from its patch Dense onwards, on caller-supplied patch rows [b, np, patch_dim] instead of an image (T2T-ViT's
        tokenizer output, t2t.py:74,100-115).  `backward(dlogits, want_dimg=True)` afterwards returns d(patches) as its second result."""
        x, proto = self._as_host(patches)
        assert x.ndim == 3, "expected patch rows [b, np, patch_dim]"
        b, n, pd = x.shape
        want = self._cfg.patch_h * self._cfg.patch_w * self._cfg.channels     # the C side copies b * np * patch_dim floats from this pointer
        assert pd == want, f"expected patches of {want} features (patch_h * patch_w * channels), got {pd}"
        h = self._ensure_handle(b)
        self._img_shape = (b, n, pd)
        out = np.empty((b, self.num_classes), dtype=np.float32)
        seed = int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed)
        N.check(N.lib().vitx_forward_patches(h, x.ctypes.data_as(C.c_void_p), b, n, 1 if training else 0, seed, out.ctypes.data_as(C.c_void_p)))
        return self._like(out, proto)

    def backward(self, dlogits, want_dimg: bool = False):
        """VJP for the last forward (what tf.GradientTape.gradient would return, README.md:746-749).
        Returns ({name: grad}, dimg|None)."""
        if self._handle is None:
            raise N.VitxError(N.ERR_STATE, "backward requires a preceding forward")
        d, _ = self._as_host(dlogits)
        dimg = None
        dimg_p = None
        if want_dimg:
            dimg = np.empty(self._last_img_shape(), dtype=np.float32)
            dimg_p = dimg.ctypes.data_as(C.c_void_p)
        N.check(N.lib().vitx_backward(self._handle, d.ctypes.data_as(C.c_void_p), dimg_p))
        self._finish_exchange()   # data parallel: the gradients every rank sees are the mean over the ranks (comm_init below)
        g = np.empty(self._n, dtype=np.float32)
        N.check(N.lib().vitx_get_grads(self._handle, g.ctypes.data_as(C.c_void_p), self._n))
        grads = {n: g[o:o + int(np.prod(s))].reshape(s) for n, s, o in self._table}
        return grads, dimg

    def _finish_exchange(self) -> None:
        """ EVERY Python backward path on this handle ends here: after comm_init a backward pass launches bucket collectives from inside
        the engine, and vitx_allreduce_grads must send the rest and join before the gradients are read (one call per backward)."""
        if getattr(self, "_comm_world", 0):
            N.check(N.lib().vitx_allreduce_grads(self._handle))

    def _refuse_exchange(self, what: str) -> None:
        """Wrapper objects (MAE / SimMIM / MPP / DistillWrapper) keep variables of their own OUTSIDE this handle's gradient arena: the native
        exchange would average the encoder's gradients and leave the wrapper's local.  Refused instead of half-done."""
        if getattr(self, "_comm_world", 0):
            raise N.VitxError(N.ERR_STATE, f"{what}: this model has joined an RCCL group (comm_init), but the wrapper's own variables live outside its "
                                           "gradient arena and would not be exchanged; use vit_tensorflow.parallel.GradSync on the wrapper's gradients instead")

    def comm_destroy(self) -> None:
        """Leave the RCCL group (communicator, communication stream, wire buffer); backward() returns local gradients again."""
        if self._handle is not None:   # (also after a comm_init that failed half way: a communicator without the overlap state)
            N.check(N.lib().vitx_comm_destroy(self._handle))
        self._comm_world = 0

    # ---- data parallel (no reference counterpart): one process per GPU, parameters replicated, images sharded, ONE exchange per step --
    # the mean of the gradient arena over the ranks, done by the library itself over RCCL (csrc/comm.hip; no torch.distributed involved)
    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte RCCL id: make it on rank 0 and hand the same bytes to every rank's comm_init (file, socket, MPI, torch store ...)."""
        uid = (C.c_char * 128)()
        N.check(N.lib().vitx_comm_unique_id(uid))
        return bytes(uid)

    def comm_init(self, rank: int, world: int, unique_id: bytes, overlap: bool = True, bucket_mb: float = 32.0, wire: str = "fp32") -> None:
        """Join the RCCL group.  From then on backward() returns the gradients averaged over the ranks (feed every rank its own shard of the
        batch, dlogits scaled by 1/local batch).  overlap: buckets of `bucket_mb` MiB are all-reduced on the handle's communication stream
        while the rest of the backward pass runs; wire "bf16" halves the bytes on xGMI (each rank's addend rounded once)."""
        if self._handle is None:
            raise N.VitxError(N.ERR_STATE, "comm_init needs a built model (call build() or run a forward first)")
        assert wire in ("fp32", "bf16") and len(unique_id) == 128
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        N.check(N.lib().vitx_comm_init(self._handle, int(rank), int(world), buf))
        N.check(N.lib().vitx_comm_overlap(self._handle, 1 if overlap else 0, int(bucket_mb * (1 << 20)), 1 if wire == "bf16" else 0))
        self._comm_world = int(world)

    def apply_gradients(self, optimizer: str = "adamw", lr: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-7,
                        weight_decay: float = 0.0, momentum: float = 0.0) -> None:
        """One optimizer step on the device with the gradients of the last backward (the reference ships no optimizer;
        eps defaults to Keras' 1e-7)."""
        if self._handle is None:
            raise N.VitxError(N.ERR_STATE, "apply_gradients requires forward + backward first")
        if optimizer == "adamw":
            N.check(N.lib().vitx_adamw_step(self._handle, lr, beta1, beta2, eps, weight_decay))
        elif optimizer == "sgd":
            N.check(N.lib().vitx_sgd_step(self._handle, lr, momentum, weight_decay))
        else:
            raise ValueError("optimizer must be 'adamw' or 'sgd'")
        self._device_newer = True

    def _last_img_shape(self):
        return self._img_shape

    def _transformer_call(self, tokens, training=True, seed=None):
        """Transformer.call(x, training=True) (vit.py:99-104): with training the Dropout layers inside the blocks are active
        (masks from `seed`, drawn when None), exactly as in the full forward."""
        x, proto = self._as_host(tokens)
        assert x.ndim == 3 and x.shape[2] == self.dim, "expected tokens [b, n, dim]"
        b, n, _ = x.shape
        h = self._ensure_handle(b)
        out = np.empty_like(x)
        seed = int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed)
        N.check(N.lib().vitx_transformer_forward(h, x.ctypes.data_as(C.c_void_p), b, n, 1 if training else 0, seed,
                                                 out.ctypes.data_as(C.c_void_p)))
        return self._like(out, proto)

    def _transformer_backward(self, dout, want_dtokens=True):
        if self._handle is None:
            raise N.VitxError(N.ERR_STATE, "transformer.backward requires a preceding transformer(tokens) call")
        d, proto = self._as_host(dout)
        assert d.ndim == 3 and d.shape[2] == self.dim, "expected d(out) [b, n, dim]"
        dtok = np.empty_like(d) if want_dtokens else None
        N.check(N.lib().vitx_transformer_backward(self._handle, d.ctypes.data_as(C.c_void_p),
                                                  dtok.ctypes.data_as(C.c_void_p) if want_dtokens else None))
        self._finish_exchange()
        g = np.empty(self._n, dtype=np.float32)
        N.check(N.lib().vitx_get_grads(self._handle, g.ctypes.data_as(C.c_void_p), self._n))
        grads = {n: g[o:o + int(np.prod(s))].reshape(s) for n, s, o in self._table}
        return grads, (self._like(dtok, proto) if want_dtokens else None)

    def debug_read(self, which: str, layer: int = 0) -> np.ndarray:
        n = C.c_int64()
        N.check(N.lib().vitx_debug_read(self._handle, which.encode(), layer, None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        N.check(N.lib().vitx_debug_read(self._handle, which.encode(), layer, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out
