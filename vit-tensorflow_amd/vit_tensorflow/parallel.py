"""Batch-sharded data parallelism for the hot path: one process per GPU, parameters replicated,
images sharded, ONE exchange step per training step -- an all-reduce (sum, then x 1/world) of the
gradient arena over RCCL/xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

The reference has no distributed code; semantics are defined by equivalence:
N ranks x b/N images with mean-reduced gradients == one device on the concatenated batch.

Buckets are fixed contiguous slices of the gradient arena.  The engine reports arena ranges as soon as
every kernel producing them has been enqueued (vitx_set_grad_ready_callback); a bucket is all-reduced
asynchronously the moment it is fully covered, so the collective overlaps the rest of backward.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are tens of MB so that RCCL can spread
each one over all links/channels instead of paying per-message latency.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradSync:
    """Bucketed asynchronous all-reduce of the flat fp32 gradient arena.

    wire_dtype=torch.bfloat16 sends each bucket as bf16 (half the bytes over xGMI: 173 MB instead of 346 MB for ViT-B, 608 MB
    instead of 1217 MB for ViT-L): the bucket is rounded to bf16 on the engine's stream, summed in bf16 by the collective and
    widened back into the fp32 arena when the step finishes.  The rounding (2^-9 relative per addend) is the usual price of
    compressed gradient exchange; the default keeps fp32 on the wire."""

    def __init__(self, grads: torch.Tensor, bucket_elems: int = 8 << 20, group=None, average: bool = True, always_reduce: bool = False,
                 wire_dtype: Optional[torch.dtype] = None):
        assert grads.dim() == 1 and grads.is_contiguous()
        assert wire_dtype in (None, torch.float32, torch.bfloat16), "wire_dtype must be None / float32 / bfloat16"
        self.wire = None if wire_dtype in (None, torch.float32) else wire_dtype
        self._wire_buf = torch.empty(grads.numel(), dtype=self.wire, device=grads.device) if self.wire is not None else None
        self._wired: List[int] = []
        self.grads = grads
        self.n = grads.numel()
        self.bucket = max(1, int(bucket_elems))
        self.nb = (self.n + self.bucket - 1) // self.bucket
        self.group = group
        self.average = average
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.always_reduce = always_reduce and dist.is_initialized()   # issue the collective even for a group of one (plumbing tests)
        self._covered = [0] * self.nb
        self._launched = [False] * self.nb
        self._works: List = []

    def _size(self, i: int) -> int:
        return min(self.bucket, self.n - i * self.bucket)

    def _launch(self, i: int) -> None:
        if self._launched[i]:
            return
        self._launched[i] = True
        if self.world == 1 and not self.always_reduce:
            return
        sl = self.grads[i * self.bucket: i * self.bucket + self._size(i)]
        if self.wire is not None:
            wb = self._wire_buf[i * self.bucket: i * self.bucket + self._size(i)]
            wb.copy_(sl)                      # fp32 -> bf16 on the current stream, ordered after the kernels that produced the bucket
            self._wired.append(i)
            sl = wb
        self._works.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def begin(self) -> None:
        self._covered = [0] * self.nb
        self._launched = [False] * self.nb
        self._works = []
        self._wired = []

    def on_ready(self, offset: int, count: int) -> None:
        """Arena range [offset, offset+count) is final (all producing kernels enqueued on the current stream)."""
        lo, hi = max(0, offset), min(self.n, offset + count)
        if hi <= lo:
            return
        for i in range(lo // self.bucket, (hi - 1) // self.bucket + 1):
            b0, b1 = i * self.bucket, i * self.bucket + self._size(i)
            self._covered[i] += max(0, min(hi, b1) - max(lo, b0))
            if self._covered[i] >= self._size(i):
                self._launch(i)

    def finish(self) -> None:
        """All-reduce whatever was not reported, wait, and turn the sum into the mean."""
        for i in range(self.nb):
            self._launch(i)
        for w in self._works:
            w.wait()
        self._works = []
        for i in self._wired:             # widen the reduced bf16 buckets back into the fp32 arena
            lo, hi = i * self.bucket, i * self.bucket + self._size(i)
            self.grads[lo:hi].copy_(self._wire_buf[lo:hi])
        self._wired = []
        if self.average and self.world > 1:
            self.grads.mul_(1.0 / self.world)


def init_from_env(backend: Optional[str] = None, force: bool = False) -> tuple:
    """(rank, local_rank, world) from the torchrun environment; initialises the default process group
    when WORLD_SIZE > 1 (or when `force` asks for a group of one)."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        be = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if be == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=be, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=be, rank=rank, world_size=world)
    return rank, local, world


def shard_range(global_batch: int, rank: int, world: int) -> range:
    """rank r takes images [r*b_local, (r+1)*b_local)  (SURVEY.md section 8e)."""
    assert global_batch % world == 0, "global batch must divide evenly over the ranks"
    b = global_batch // world
    return range(rank * b, (rank + 1) * b)


def broadcast_params(params: torch.Tensor, src: int = 0, group=None) -> None:
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(params, src=src, group=group)
