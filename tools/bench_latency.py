"""Small-batch step latency, eager launches vs one HIP-graph launch (vitx_graph_*), on the README configuration
(`ViT(image_size=256, patch_size=32, dim=1024, depth=6, heads=16, mlp_dim=2048)`, README.md:49-59) or any bench workload.

    python tools/bench_latency.py [workload] [batch] [steps]       -> one JSON line"""
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vit-tensorflow_amd")]
from vit_tensorflow import ViT, _native as N   # noqa: E402

WORKLOADS = {
    "vit_readme_256": dict(image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=6, heads=16, mlp_dim=2048),
    "vit_b16_224": dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072),
    # parallel_vit.py:177-188 (the reference's usage example: 257 tokens, two parallel branches per layer)
    "parallel_vit_readme": dict(image_size=256, patch_size=16, num_classes=1000, dim=1024, depth=6, heads=8, mlp_dim=2048, num_parallel_branches=2),
}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vit_readme_256"
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    kw = WORKLOADS[name]
    if "num_parallel_branches" in kw:
        from vit_tensorflow.parallel_vit import ViT as Model
    else:
        Model = ViT
    m = Model(**kw, compute="bf16", max_batch=b, seed=0)
    h = m._ensure_handle(b)
    lib = N.lib()
    dev = torch.device("cuda:0")
    S = kw["image_size"]
    img = torch.randn(b, S, S, 3, device=dev)
    labels = torch.randint(0, kw["num_classes"], (b,), device=dev, dtype=torch.int32)
    torch.cuda.synchronize()

    def step():
        N.check(lib.vitx_params_changed(h))
        N.check(lib.vitx_forward_dev(h, C.c_void_p(img.data_ptr()), b, S, S, 0, 0, None))
        N.check(lib.vitx_ce_loss_grad_dev(h, C.c_void_p(labels.data_ptr()), 1.0 / b, None))
        N.check(lib.vitx_backward_dev(h, None, None))
        N.check(lib.vitx_sgd_step(h, 1e-3, 0.0, 0.0))

    def timed(fn):
        for _ in range(5):
            fn()
        N.check(lib.vitx_sync(h))
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        N.check(lib.vitx_sync(h))
        return 1e3 * (time.perf_counter() - t0) / steps

    eager = timed(step)
    N.check(lib.vitx_graph_capture_begin(h))
    step()
    g = C.c_void_p()
    N.check(lib.vitx_graph_capture_end(h, C.byref(g)))
    graph = timed(lambda: N.check(lib.vitx_graph_launch(h, g)))
    N.check(lib.vitx_graph_destroy(g))
    print(json.dumps({"workload": f"{name} training step (refresh + fwd + CE grad + bwd + SGD), batch {b}, bf16", "eager_ms_per_step": round(eager, 4),
                      "graph_ms_per_step": round(graph, 4), "speedup": round(eager / graph, 3), "steps": steps}))


if __name__ == "__main__":
    main()
