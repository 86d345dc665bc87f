#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): images/sec, forward+backward, ViT-B/16 224 px, bf16 operands,
batch 256 per GPU, synthetic images resident in HBM, one process per GPU (weak scaling).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = re-derive the bf16 operand copies of the (possibly updated) fp32 weights -> forward ->
softmax-CE gradient -> backward of every parameter [-> bucketed RCCL all-reduce of the gradients].
Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` for the dominant
kernel class (the bf16 MFMA GEMM, timed with HIP events on the engine's stream) and `cpu_baseline`
(the oracle's torch-CPU fp32 restatement of vit.py, timed on this box's host cores; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vit-tensorflow_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_BF16_PEAK = 2516.6e12   # 256 CU x 4096 FLOP/clk/CU x 2.4 GHz (MI355X_MICROARCH.md: ~2.5 PF dense bf16)
HBM_PEAK = 8.0e12

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "vit_b16_224": dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072),
    # configs[2] per-GPU shape (ViT-L/16)
    "vit_l16_224": dict(image_size=224, patch_size=16, num_classes=1000, dim=1024, depth=24, heads=16, mlp_dim=4096),
    # north_star's "256x256x3 synthetic images" wording applied to ViT-B/16: 257 tokens (SURVEY.md section 8 header)
    "vit_b16_256": dict(image_size=256, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072),
    # configs[0] (README example; plumbing-sized)
    "vit_readme_256": dict(image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=6, heads=16, mlp_dim=2048),
    # configs[3]: DeepViT with Re-attention; configs[4]: CaiT (LayerScale + talking heads + class attention)
    "deepvit_256": dict(variant="deepvit", image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=12, heads=16, mlp_dim=2048),
    "cait_256": dict(variant="cait", image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=24, cls_depth=2, heads=16, mlp_dim=2048),
}


def flops_per_image(kw) -> float:
    from vit_tensorflow._model_math import flops_per_image as f   # the package's own closed form
    kw = dict(kw)
    return f(kw.pop("variant", "vit"), **kw)


def cpu_baseline(kw, seconds: float, batch: int = 16):
    """Reference-restatement CPU baseline (torch-CPU fp32, NOT TensorFlow: TF is absent from the image): the oracle's op-for-op
    restatement of vit.py, forward + autograd backward, batch 16, on the thread count that measures fastest on this host
    (torch-CPU with hundreds of threads on a shared host oversubscribes badly; 16 / 32 / 64 are tried for one step each), then
    AT LEAST FIVE timed steps (more until the time budget is used)."""
    from oracle import ref_torch, spec
    cfg = spec.make_config("vit", **kw)
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    P = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in spec.init_params(cfg, 1).items()}
    img = torch.randn(batch, *cfg["image_size"], 3)
    labels = torch.randint(0, cfg["num_classes"], (batch,))

    def step():
        for p in P.values():
            p.grad = None
        logits = ref_torch.forward(cfg, P, img)
        torch.nn.functional.cross_entropy(logits, labels).backward()

    forced = os.environ.get("VITX_CPU_BASELINE_THREADS")
    cands = [int(forced)] if forced else sorted({min(avail, t) for t in (16, 32, 64)})
    torch.set_num_threads(cands[0])
    step()   # warm-up (allocator, thread pool)
    tried = {}
    for t in cands:
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        step()
        tried[t] = time.perf_counter() - t0
    ncores = min(tried, key=tried.get)
    torch.set_num_threads(ncores)
    budget = max(0.0, seconds - sum(tried.values()))
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if (el >= budget and n >= 5) or n >= 50:
            break
    return {"value": round(batch * n / el, 3), "unit": "images/sec", "cores": ncores, "host_cpus": avail, "kind": "port",
            "threads_tried_s_per_step": {str(k): round(v, 2) for k, v in tried.items()},
            "sample": f"oracle/ref_torch.py (unfused torch-CPU fp32 restatement of vit.py, autograd backward), ViT-B/16 224 "
                      f"batch {batch}, {n} fwd+bwd steps in {el:.1f} s on {ncores} threads; TensorFlow itself is not installable here"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)    # ~2 s of GPU time at batch 256: long enough for a utilisation sampler to see it
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--workload", default="vit_b16_224", choices=sorted(WORKLOADS))
    ap.add_argument("--compute", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step as one HIP graph (single GPU; the default launches kernel by kernel)")
    ap.add_argument("--bucket-mb", type=float, default=48.0)
    ap.add_argument("--grad-wire", default="fp32", choices=["fp32", "bf16"], help="dtype of the gradient all-reduce on the wire (N > 1)")
    ap.add_argument("--dp-impl", default="native", choices=["native", "torch"],
                    help="gradient exchange (N > 1): native = the library's own bucketed RCCL exchange on its communication stream (csrc/comm.hip); "
                         "torch = torch.distributed all-reduce per bucket from the gradient-ready callback (vit_tensorflow/parallel.py)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Native libraries (RCCL prints a version banner through C stdio, flushed at exit)
    # would add lines of their own, so fd 1 is pointed at stderr for the whole run and the result is written to the saved fd.
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from vit_tensorflow import ViT, _native as N
    from vit_tensorflow.parallel import GradSync, broadcast_params, init_from_env
    import torch.distributed as dist

    force_dp = bool(os.environ.get("VITX_FORCE_DP"))   # exercise the whole DP path (RCCL group of 1) on a single GPU
    native_dp = args.dp_impl == "native"
    gloo_default = native_dp   # the default process group carries CPU tensors only (rendezvous, barrier, max of the timing)
    # native exchange: torch.distributed only carries the rendezvous (RCCL id, initial weights), the barrier and the max over ranks -- gloo on the CPU
    # VITX_BENCH_ONE_GPU=1 (tests/test_gpu_bench_ranks.py): every rank on GPU 0 -- the N > 1 code path of this file (rendezvous, weight broadcast, exchange,
    # barrier, max over ranks, one JSON line) on a one-GPU box.  RCCL refuses two ranks on one device: the native exchange then needs VITX_RCCL_LIB
    # (tests/fake_rccl), the torch exchange runs its buckets over gloo.  Never set for a measurement.
    one_gpu = bool(os.environ.get("VITX_BENCH_ONE_GPU"))
    rank, local, world = init_from_env(backend="gloo" if (native_dp or one_gpu) else None, force=force_dp)
    if one_gpu:
        local = 0
        gloo_default = True
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path is HIP-only (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    kw = dict(WORKLOADS[args.workload])
    variant = kw.pop("variant", "vit")
    b = args.batch
    lib = N.lib()
    if variant == "deepvit":
        from vit_tensorflow.deepvit import DeepViT as Model
    elif variant == "cait":
        from vit_tensorflow.cait import CaiT as Model
    else:
        Model = ViT
    # torch exchange: the stream the engine will run on is created (and used once) BEFORE the handle makes its own streams.  Hardware queues are spread
    # over the GPU's four command-processor pipes in creation order; a compute stream made later lands on the pipe of the weight-gradient or
    # small-reduction stream by luck, and two busy queues on one pipe serialise their dispatches
    dp_stream = None
    if (world > 1 or force_dp) and not native_dp:
        dp_stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(dp_stream):
            torch.zeros(1, device=dev)
        torch.cuda.synchronize()
    model = Model(**kw, compute=args.compute, max_batch=b, device=local, seed=1)
    model.build((b,))
    h = model._handle
    H, W = (kw["image_size"],) * 2
    # synthetic inputs, resident in HBM before the timed region (tf.random.normal analogue, README.md:61)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    img = torch.randn(b, H, W, 3, device=dev, generator=g)
    labels = torch.randint(0, kw["num_classes"], (b,), device=dev, generator=g, dtype=torch.int32)
    torch.cuda.synchronize()

    sync = None
    cb = None
    dp_group = None   # torch exchange: the default group (RCCL), or an NCCL group of its own when it is the fallback of the native exchange (default group = gloo)
    dp = world > 1 or force_dp
    if dp and native_dp:
        # every rank starts from rank 0's weights (the packed host blob; one-time, over gloo), joins the RCCL group, and from then on the library
        # exchanges the gradient buckets itself while the backward pass runs
        blob = np.empty(model._n, dtype=np.float32)
        N.check(lib.vitx_get_params(h, blob.ctypes.data_as(C.c_void_p), model._n))
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(model.comm_unique_id()), dtype=torch.uint8).clone()
        if world > 1:
            wt = torch.from_numpy(blob)
            dist.broadcast(wt, src=0)
            dist.broadcast(uid, src=0)
            N.check(lib.vitx_set_params(h, blob.ctypes.data_as(C.c_void_p), model._n))
        # a rank whose librccl cannot be opened or initialised reports an error here (a hang inside RCCL cannot be caught): every rank then
        # agrees, over gloo, to fall back to the torch exchange on an NCCL group of its own instead of leaving the run without a number
        ok = 1
        try:
            if os.environ.get("VITX_BENCH_SIMULATE_NATIVE_FAILURE"):   # tests: exercise the fallback below
                raise RuntimeError("simulated")
            model.comm_init(rank, world, bytes(uid.numpy().tobytes()), overlap=True, bucket_mb=args.bucket_mb, wire=args.grad_wire)
        except Exception as ex:   # noqa: BLE001
            ok = 0
            print(f"[bench] rank {rank}: native exchange unavailable ({ex}); falling back to --dp-impl torch", file=sys.stderr)
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            native_dp = False
            #  a rank that DID get its communicator must give it up before the torch exchange takes over: left alive, its overlapped
            # buckets would still be launched from inside the backward pass and wait for peers that never join
            try:
                model.comm_destroy()
            except Exception as ex:   # noqa: BLE001
                print(f"[bench] rank {rank}: comm_destroy: {ex}", file=sys.stderr)
            dp_group = dist.new_group(backend="nccl") if dist.is_initialized() else None
    if dp and not native_dp:
        # The engine must run on the stream RCCL orders itself against.  torch's default stream has handle 0, which the C ABI
        # reads as "use the library's own stream", so the data-parallel path runs under an explicit side stream.
        if dp_stream is None:   # (the fall-back from a native exchange that could not start: created late, wherever it lands)
            dp_stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(dp_stream)
        n = C.c_int64()
        p = C.c_void_p()
        N.check(lib.vitx_params_dev(h, C.byref(p), C.byref(n)))
        params_t = torch.empty(n.value, device=dev, dtype=torch.float32)
        grads_t = torch.zeros(n.value, device=dev, dtype=torch.float32)
        N.check(lib.vitx_bind_arenas(h, C.c_void_p(params_t.data_ptr()), C.c_void_p(grads_t.data_ptr())))
        # engine work must be ordered with RCCL through torch's current stream
        N.check(lib.vitx_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        broadcast_params(params_t, 0, group=dp_group)
        N.check(lib.vitx_params_changed(h))
        sync = GradSync(grads_t, bucket_elems=int(args.bucket_mb * (1 << 20) / 4), group=dp_group, average=False, always_reduce=force_dp,
                        wire_dtype=torch.bfloat16 if args.grad_wire == "bf16" else None)
        cb = N.GRAD_READY_FN(lambda _u, off, cnt: sync.on_ready(int(off), int(cnt)))
        N.check(lib.vitx_set_grad_ready_callback(h, cb, None))
    # torch exchange: dlogits carry 1/global_batch and the all-reduce is a plain sum; native exchange: vitx_allreduce_grads AVERAGES over the
    # ranks (sum x 1/world), so each rank's dlogits carry 1/local batch -- the same gradient of the global mean loss either way
    inv_global = 1.0 / float(b) if (dp and not sync) else 1.0 / float(b * world)

    def step():
        N.check(lib.vitx_params_changed(h))    # a training step sees updated fp32 weights: re-derive bf16 operands
        if sync:
            sync.begin()
        N.check(lib.vitx_forward_dev(h, C.c_void_p(img.data_ptr()), b, H, W, 0, 0, None))
        N.check(lib.vitx_ce_loss_grad_dev(h, C.c_void_p(labels.data_ptr()), inv_global, None))
        N.check(lib.vitx_backward_dev(h, None, None))
        if sync:
            sync.finish()
        elif dp:
            N.check(lib.vitx_allreduce_grads(h))   # sends what the backward pass has not, orders the compute stream behind the last bucket

    def full_sync():
        N.check(lib.vitx_sync(h))
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()

    # The first step of a process also measures the GEMM tile variants of every shape it meets (a few extra launches per shape) and
    # makes the first-use allocations; with --warmup 0 that step would land inside the timed region, so one untimed priming step
    # runs in that case (reported as "priming_steps"; "warmup" stays what was asked for).
    priming = 1 if args.warmup == 0 else 0
    for _ in range(args.warmup + priming):
        step()
    full_sync()
    run_step = step
    if args.graph and not dp:
        # vitx_graph_*: the whole step (operand refresh, forward, CE gradient, backward) recorded once and replayed as ONE launch
        N.check(lib.vitx_graph_capture_begin(h))
        step()
        graph = C.c_void_p()
        N.check(lib.vitx_graph_capture_end(h, C.byref(graph)))
        run_step = lambda: N.check(lib.vitx_graph_launch(h, graph))
        for _ in range(2):
            run_step()
        full_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    full_sync()
    el = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([el], device=torch.device("cpu") if gloo_default else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    fpi = flops_per_image(WORKLOADS[args.workload])
    value = b * world * args.steps / el
    out = {
        "metric": "images/sec (fwd+bwd) ViT-B/16 224px bf16" if args.workload == "vit_b16_224" else f"images/sec (fwd+bwd) {args.workload}",
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * el / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16": "bf16", "fp32": "f32", "bf16x3": "bf16x3 (fp32 storage, split-operand bf16 MFMA GEMMs)"}[args.compute], "data": "synthetic",
        "config": {"workload": f"{args.workload} fwd+bwd, batch {b}/GPU, N(0,1) NHWC images resident in HBM, random-init weights, "
                               f"softmax-CE cotangent, dropout 0", "global_batch": b * world,
                   "parallelism": f"dp{world}", "compute": args.compute, **({"launch": "hip_graph"} if (args.graph and not dp) else {}), **({"grad_wire": args.grad_wire, "dp_impl": args.dp_impl if (native_dp or args.dp_impl == "torch") else "torch (native exchange unavailable)"} if dp else {})},
        "path_mfma_frac": round(value / world * fpi / MFMA_BF16_PEAK, 4),
        "flops_per_image": fpi,
    }
    if priming:
        out["priming_steps"] = priming
    if dp and native_dp:
        st = (C.c_int64 * 4)()
        N.check(lib.vitx_comm_stats(h, st))
        out["config"]["exchange"] = {"buckets": int(st[0]), "sent_during_backward": int(st[1]), "bucket_mib": round(st[2] * 4 / (1 << 20), 1),
                                     "dense_launches_beside_a_collective": int(st[3])}

    if rank == 0 and not args.no_profile:
        try:   # per-kernel-class HIP-event timing of two more steps (outside the timed region)
            if sync:
                N.check(lib.vitx_set_grad_ready_callback(h, C.cast(None, N.GRAD_READY_FN), None))
            elif dp:
                N.check(lib.vitx_comm_overlap(h, 0, 0, 0))   # the per-class timing below is of the compute kernels alone
            psteps = 2
            N.check(lib.vitx_profile_begin(h))
            for _ in range(psteps):
                N.check(lib.vitx_params_changed(h))
                N.check(lib.vitx_forward_dev(h, C.c_void_p(img.data_ptr()), b, H, W, 0, 0, None))
                N.check(lib.vitx_ce_loss_grad_dev(h, C.c_void_p(labels.data_ptr()), inv_global, None))
                N.check(lib.vitx_backward_dev(h, None, None))
            stats = (N.KernelStat * 160)()
            ns = C.c_int32()
            N.check(lib.vitx_profile_end(h, stats, 160, C.byref(ns)))
            classes, shapes = {}, []
            epi_names = {0: "bf16 store", 1: "fp32 store", 2: "bias+GELU (act, gelu')", 3: "bias+fp32 residual", 4: "patch embed",
                         5: "x gelu' + column sums", 6: "split-K partials"}
            nstat = min(ns.value, 160)   # (vitx_profile_end reports the number of classes even when it exceeds the capacity passed in)
            for i in range(nstat):
                s = stats[i]
                name = s.name.decode()
                if name.startswith("shape "):
                    # per-shape rows of the dominant family ("shape nt e<epilogue> MxNxK" / "shape tn s<slices> MxNxK"; the same launches
                    # are also booked under gemm_bf16_mfma / gemm_bf16_mfma_tn)
                    _, form, tag, dims = name.split()
                    M_, N_, K_ = (int(v) for v in dims.split("x"))
                    us = 1e3 * s.total_ms / max(s.launches, 1)
                    shapes.append({"form": form, "M": M_, "N": N_, "K": K_,
                                   "epilogue": epi_names.get(int(tag[1:]), tag) if form == "nt" else f"fp32 partials, {tag[1:]} K slices",
                                   "launches_per_step": s.launches // psteps, "us": round(us, 1),
                                   "tflops": round(s.flops / max(s.total_ms, 1e-9) / 1e9, 1), "frac": round(s.flops / max(s.total_ms, 1e-9) / 1e-3 / MFMA_BF16_PEAK, 3)})
                    continue
                classes[name] = {"launches_per_step": s.launches // psteps, "ms_per_step": round(s.total_ms / psteps, 4),
                                 "tflops": round(s.flops / max(s.total_ms, 1e-9) / 1e9, 1) if s.flops else None,
                                 "gbps": round(s.bytes / max(s.total_ms, 1e-9) / 1e6, 1) if s.bytes else None}
            out["kernel_classes"] = classes
            if shapes:
                out["gemm_shapes"] = sorted(shapes, key=lambda r: -r["us"] * r["launches_per_step"])
            # dominant kernel family = the bf16 MFMA GEMM (NT form for forward/dgrad, TN form for the weight gradients)
            fam = [stats[i] for i in range(nstat) if stats[i].name.decode().startswith("gemm_bf16_mfma")]
            peak = MFMA_BF16_PEAK
            if not fam and args.compute == "bf16x3":   # split-operand mode: three bf16 MFMA products per fp32 product (achieved = fp32-equivalent FLOPs)
                fam = [stats[i] for i in range(nstat) if stats[i].name.decode() == "gemm_bf16x3_mfma"]
                peak = MFMA_BF16_PEAK / 3.0
            if not fam:   # fp32 parity mode: the exact-fp32 matrix-pipe kernel (+ the scalar kernel for small / strided problems)
                fam = [stats[i] for i in range(nstat) if stats[i].name.decode() in ("gemm_f32_mfma", "gemm_generic_fma")]
                peak = 157.3e12
            if fam:
                fl = sum(s.flops for s in fam); ms = sum(s.total_ms for s in fam); ln = sum(s.launches for s in fam)
                ach = fl / (ms * 1e-3)
                # HBM traffic of the family per launch: PMC counters cannot be collected from inside this process; the committed
                # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE reduction of this very command (tools/pmc_traffic.py) is reported
                # -- only when that profile was taken on the kernel sources this run is built from (it carries their digest)
                traffic, traffic_src = None, None
                try:
                    from vit_tensorflow._model_math import kernel_source_id
                    tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
                    if args.workload == "vit_b16_224" and b == 256 and os.path.exists(tp):
                        with open(tp) as f:
                            prof = json.load(f)
                        if prof.get("kernel_source_id") == kernel_source_id():
                            traffic = prof.get("gemm_family_hbm_bytes_per_launch")
                            traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, same kernel sources)"
                        else:
                            traffic_src = (f"profiles/pmc_traffic.json is from other kernel sources ({prof.get('kernel_source_id')} vs "
                                           f"{kernel_source_id()}): not quoted; re-run tools/gpu_round.sh <tag> pmc_bench")
                except Exception:
                    traffic = None
                out["roofline"] = {"bound": "mfma", "kernel": "+".join(s.name.decode() for s in fam), "achieved": round(ach / 1e12, 2),
                                   "peak": round(peak / 1e12, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                                   "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_src,
                                   "algorithmic_bytes_per_launch": round(sum(s.bytes for s in fam) / max(ln, 1)),
                                   "launches_per_step": int(ln // psteps), "avg_launch_ms": round(ms / ln, 5),
                                   "algorithmic_tflop_per_step": round(fl / psteps / 1e12, 4)}
        except Exception as ex:   # never lose the headline line to a diagnostics problem
            out["roofline_error"] = repr(ex)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(WORKLOADS["vit_b16_224"], args.cpu_seconds)
        except Exception as ex:
            out["cpu_baseline_error"] = repr(ex)

    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
