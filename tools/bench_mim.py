"""Time one MAE / SimMIM pre-training step (forward + backward, inputs resident in HBM) on ViT-B/16 224.

    python tools/bench_mim.py [mae|simmim] [batch] [steps]

Prints one JSON line.  The wrappers' index / masking kernels and the two small Dense layers run next to the encoder's
ordinary kernels; this script is how DESIGN.md's wrapper numbers were measured."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-tensorflow_amd"))
from vit_tensorflow import ViT, MAE, SimMIM, _native as N   # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "mae"
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    enc = ViT(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072, compute="bf16", max_batch=b, seed=0)
    if kind == "mae":
        w = MAE(image_size=224, encoder=enc, decoder_dim=512, masking_ratio=0.75, decoder_depth=6, decoder_heads=8, decoder_dim_head=64,
                literal_loss=False, seed=1)
    else:
        w = SimMIM(image_size=224, encoder=enc, masking_ratio=0.5, seed=1)
    m = w._ensure(b)
    lib = N.lib()
    dev = torch.device("cuda:0")
    img = torch.randn(b, 224, 224, 3, device=dev)
    npat, nm = w.num_masked()
    idx_host = w._draw_indices(b, npat, nm)
    idx = torch.from_numpy(np.ascontiguousarray(idx_host)).to(dev)
    loss = torch.zeros(1, device=dev)
    torch.cuda.synchronize()

    def step():
        N.check(lib.vitx_params_changed(enc._handle))
        if w.decoder is not None:
            N.check(lib.vitx_params_changed(w.decoder._handle))
        N.check(lib.vitx_mim_params_changed(m))
        N.check(lib.vitx_mim_forward_dev(m, C.c_void_p(img.data_ptr()), b, 224, 224, C.c_void_p(idx.data_ptr()), 1, 0, C.c_void_p(loss.data_ptr())))
        N.check(lib.vitx_mim_backward(m))

    for _ in range(3):
        step()
    N.check(lib.vitx_sync(enc._handle))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    N.check(lib.vitx_sync(enc._handle))
    el = time.perf_counter() - t0
    out = {"workload": f"{kind} pre-training step, ViT-B/16 224 encoder, batch {b}, bf16", "ms_per_step": round(1e3 * el / steps, 3),
           "images_per_sec": round(b * steps / el, 1), "num_patches": npat, "num_masked": nm, "loss": float(loss.item())}
    if os.environ.get("MIM_PROFILE"):
        N.check(lib.vitx_profile_begin(enc._handle))
        step()
        stats = (N.KernelStat * 64)()
        n = C.c_int32()
        N.check(lib.vitx_profile_end(enc._handle, stats, 64, C.byref(n)))
        out["encoder_kernel_classes_ms"] = {stats[i].name.decode(): round(stats[i].total_ms, 3) for i in range(n.value)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
