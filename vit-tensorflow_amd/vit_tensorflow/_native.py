"""ctypes binding of libvitx.so (include/vitx.h).  No compute happens in Python: every tensor op of
the hot path is a HIP kernel behind the C ABI.  There is deliberately NO CPU fallback -- if the
library is missing or no GPU is visible, calls fail loudly."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITX_LIB", os.path.join(_HERE, "..", "lib", "libvitx.so"))

VARIANT_VIT, VARIANT_DEEPVIT, VARIANT_CAIT, VARIANT_PATCH_MERGER = 0, 1, 2, 3
POOL_CLS, POOL_MEAN = 0, 1
COMPUTE_FP32, COMPUTE_BF16, COMPUTE_BF16X3 = 0, 1, 2
OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_STATE, ERR_COMM = 0, -1, -2, -3, -4, -5


class VitxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libvitx error {code}: {msg}")
        self.code = code
        self.message = msg


class Config(C.Structure):
    _fields_ = [
        ("variant", C.c_int32),
        ("image_h", C.c_int32), ("image_w", C.c_int32),
        ("patch_h", C.c_int32), ("patch_w", C.c_int32),
        ("channels", C.c_int32),
        ("num_classes", C.c_int32), ("dim", C.c_int32), ("depth", C.c_int32), ("cls_depth", C.c_int32),
        ("heads", C.c_int32), ("dim_head", C.c_int32), ("mlp_dim", C.c_int32),
        ("pool", C.c_int32),
        ("dropout", C.c_float), ("emb_dropout", C.c_float), ("layer_dropout", C.c_float),
        ("ln_eps", C.c_float),
        ("compute", C.c_int32),
        ("max_batch", C.c_int32),
        ("device_id", C.c_int32),
        ("num_parallel_branches", C.c_int32),
        ("patch_merge_layer", C.c_int32), ("patch_merge_num_tokens", C.c_int32),
        ("reserved", C.c_int32 * 5),
    ]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


MIM_MAE, MIM_SIMMIM, MIM_MPP = 0, 1, 2


class MimConfig(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("decoder_dim", C.c_int32), ("decoder_depth", C.c_int32), ("decoder_heads", C.c_int32), ("decoder_dim_head", C.c_int32),
        ("literal_loss", C.c_int32),
        ("masking_ratio", C.c_double),
        ("output_channel_bits", C.c_int32), ("max_pixel_val", C.c_float), ("has_norm", C.c_int32),
        ("norm_mean", C.c_float * 4), ("norm_std", C.c_float * 4),
        ("reserved", C.c_int32 * 5),
    ]


class DistillConfig(C.Structure):
    _fields_ = [("temperature", C.c_float), ("alpha", C.c_float), ("hard", C.c_int32), ("literal_loss", C.c_int32), ("reserved", C.c_int32 * 8)]


GRAD_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64)

# every symbol include/vitx.h declares: (name, restype, argtypes)
_P = C.POINTER
SYMBOLS: List[Tuple[str, object, list]] = [
    ("vitx_version", C.c_char_p, []),
    ("vitx_last_error", C.c_char_p, []),
    ("vitx_param_table_size", C.c_int32, [_P(Config), _P(C.c_int64), _P(C.c_int64)]),
    ("vitx_param_table_entry", C.c_int32, [_P(Config), C.c_int64, C.c_char_p, C.c_int32, _P(C.c_int64), _P(C.c_int32), _P(C.c_int64)]),
    ("vitx_create", C.c_int32, [_P(Config), _P(C.c_void_p)]),
    ("vitx_destroy", C.c_int32, [C.c_void_p]),
    ("vitx_get_config", C.c_int32, [C.c_void_p, _P(Config)]),
    ("vitx_set_params", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("vitx_get_params", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("vitx_get_grads", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("vitx_params_dev", C.c_int32, [C.c_void_p, _P(C.c_void_p), _P(C.c_int64)]),
    ("vitx_grads_dev", C.c_int32, [C.c_void_p, _P(C.c_void_p), _P(C.c_int64)]),
    ("vitx_bind_arenas", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_params_changed", C.c_int32, [C.c_void_p]),
    ("vitx_forward", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p]),
    ("vitx_forward_dev", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p]),
    ("vitx_backward", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_backward_dev", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_set_patch_input", C.c_int32, [C.c_void_p, C.c_int32]),
    ("vitx_distill_backward_input", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_forward_patches", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p]),
    ("vitx_forward_patches_dev", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p]),
    ("vitx_get_opt_state", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_set_opt_state", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    ("vitx_transformer_forward", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p]),
    ("vitx_transformer_backward", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_patch_unfold", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ("vitx_embed_forward", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ("vitx_embed_forward_dev", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ("vitx_patch_dense_forward", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    ("vitx_head_forward", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("vitx_head_forward_dev", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("vitx_head_backward", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_head_backward_dev", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_embed_backward", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_embed_backward_dev", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_extract_patches_shape", C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    ("vitx_extract_patches", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ("vitx_extract_patches_backward", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ("vitx_extract_patches_dev", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    ("vitx_extract_patches_backward_dev", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    ("vitx_ce_loss_grad_dev", C.c_int32, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    ("vitx_adamw_step", C.c_int32, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    ("vitx_sgd_step", C.c_int32, [C.c_void_p, C.c_float, C.c_float, C.c_float]),
    ("vitx_set_stream", C.c_int32, [C.c_void_p, C.c_void_p]),
    ("vitx_sync", C.c_int32, [C.c_void_p]),
    ("vitx_graph_capture_begin", C.c_int32, [C.c_void_p]),
    ("vitx_graph_capture_end", C.c_int32, [C.c_void_p, _P(C.c_void_p)]),
    ("vitx_graph_launch", C.c_int32, [C.c_void_p, C.c_void_p]),
    ("vitx_graph_destroy", C.c_int32, [C.c_void_p]),
    ("vitx_set_grad_ready_callback", C.c_int32, [C.c_void_p, GRAD_READY_FN, C.c_void_p]),
    ("vitx_comm_unique_id", C.c_int32, [C.c_void_p]),
    ("vitx_comm_init", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("vitx_comm_overlap", C.c_int32, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32]),
    ("vitx_comm_stats", C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)]),
    ("vitx_comm_destroy", C.c_int32, [C.c_void_p]),
    ("vitx_debug_switches", C.c_int32, [C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    ("vitx_allreduce_grads", C.c_int32, [C.c_void_p]),
    ("vitx_profile_begin", C.c_int32, [C.c_void_p]),
    ("vitx_profile_end", C.c_int32, [C.c_void_p, _P(KernelStat), C.c_int32, _P(C.c_int32)]),
    ("vitx_workspace_bytes", C.c_int32, [C.c_void_p, _P(C.c_int64)]),
    ("vitx_debug_read", C.c_int32, [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int64, _P(C.c_int64)]),
    ("vitx_bench_gemm", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P(C.c_float), _P(C.c_float)]),
    ("vitx_check_gemm", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P(C.c_float)]),
    ("vitx_mim_create", C.c_int32, [C.c_void_p, _P(MimConfig), _P(C.c_void_p)]),
    ("vitx_mim_destroy", C.c_int32, [C.c_void_p]),
    ("vitx_mim_decoder", C.c_void_p, [C.c_void_p]),
    ("vitx_mim_param_table_size", C.c_int32, [C.c_void_p, _P(C.c_int64), _P(C.c_int64)]),
    ("vitx_mim_param_table_entry", C.c_int32, [C.c_void_p, C.c_int64, C.c_char_p, C.c_int32, _P(C.c_int64), _P(C.c_int32), _P(C.c_int64)]),
    ("vitx_mim_set_params", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("vitx_mim_get_params", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("vitx_mim_get_grads", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("vitx_mim_params_dev", C.c_int32, [C.c_void_p, _P(C.c_void_p), _P(C.c_void_p), _P(C.c_int64)]),
    ("vitx_mim_params_changed", C.c_int32, [C.c_void_p]),
    ("vitx_mim_num_masked", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _P(C.c_int32), _P(C.c_int32)]),
    ("vitx_mim_forward", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p]),
    ("vitx_mim_forward_dev", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p]),
    ("vitx_mim_backward", C.c_int32, [C.c_void_p]),
    ("vitx_mim_read", C.c_int32, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, _P(C.c_int64)]),
    ("vitx_forward_distill", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_backward_distill", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vitx_distill_create", C.c_int32, [C.c_void_p, _P(DistillConfig), _P(C.c_void_p)]),
    ("vitx_distill_destroy", C.c_int32, [C.c_void_p]),
    ("vitx_distill_param_table_size", C.c_int32, [C.c_void_p, _P(C.c_int64), _P(C.c_int64)]),
    ("vitx_distill_param_table_entry", C.c_int32, [C.c_void_p, C.c_int64, C.c_char_p, C.c_int32, _P(C.c_int64), _P(C.c_int32), _P(C.c_int64)]),
    ("vitx_distill_set_params", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("vitx_distill_get_params", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("vitx_distill_get_grads", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("vitx_distill_forward", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_float, C.c_float, C.c_void_p]),
    ("vitx_distill_forward_dev", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_float, C.c_float, C.c_void_p]),
    ("vitx_distill_backward", C.c_int32, [C.c_void_p, C.c_void_p]),
    ("vitx_distill_read", C.c_int32, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, _P(C.c_int64)]),
]

_lib = None


def lib():
    """Load libvitx.so once.  Raises (never falls back) when the HIP library has not been built."""
    global _lib
    if _lib is None:
        path = os.path.abspath(LIB_PATH)
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: build it with `python vit-tensorflow_amd/build.py` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        l = C.CDLL(path, mode=C.RTLD_LOCAL)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)   # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        raise VitxError(rc, lib().vitx_last_error().decode("utf-8", "replace"))


def param_table(cfg: Config):
    """[(name, shape, offset)] from the C library (host-only call; no GPU needed)."""
    l = lib()
    nt, ne = C.c_int64(), C.c_int64()
    check(l.vitx_param_table_size(C.byref(cfg), C.byref(nt), C.byref(ne)))
    out = []
    name = C.create_string_buffer(256)
    shape = (C.c_int64 * 4)()
    rank, off = C.c_int32(), C.c_int64()
    for i in range(nt.value):
        check(l.vitx_param_table_entry(C.byref(cfg), i, name, 256, shape, C.byref(rank), C.byref(off)))
        out.append((name.value.decode(), tuple(int(shape[k]) for k in range(rank.value)), int(off.value)))
    return out, int(ne.value)


def mim_param_table(handle, prefix="vitx_mim"):
    """[(name, shape, offset)] of a wrapper object's own parameters (prefix 'vitx_mim': MAE / SimMIM, 'vitx_distill': DistillWrapper)."""
    l = lib()
    size_fn, entry_fn = getattr(l, prefix + "_param_table_size"), getattr(l, prefix + "_param_table_entry")
    nt, ne = C.c_int64(), C.c_int64()
    check(size_fn(handle, C.byref(nt), C.byref(ne)))
    out = []
    name = C.create_string_buffer(256)
    shape = (C.c_int64 * 4)()
    rank, off = C.c_int32(), C.c_int64()
    for i in range(nt.value):
        check(entry_fn(handle, i, name, 256, shape, C.byref(rank), C.byref(off)))
        out.append((name.value.decode(), tuple(int(shape[k]) for k in range(rank.value)), int(off.value)))
    return out, int(ne.value)


def debug_switches():
    """[(name, class, state, doc)] for every VITX_* environment variable the library reads (csrc/env.hip): class "tuning" (same results),
    "path" (another validated code path) or "diag" (may corrupt results; ignored by the release library)."""
    need = C.c_int64()
    check(lib().vitx_debug_switches(None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    check(lib().vitx_debug_switches(buf, need.value, None))
    return [tuple(line.split("\t", 3)) for line in buf.value.decode().splitlines()]
