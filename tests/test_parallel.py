"""CPU tier: the N>1 path with world_size-2 gloo.  Each rank computes the ORACLE's gradients on its
shard of the batch (the oracle stands in for the GPU engine here), the product's GradSync bucketed
all-reduce combines them, and the result must equal the single-process gradient on the whole batch
(DP equivalence)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_torch, spec
from util import oracle_cfg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(cfg, grads):
    return np.concatenate([grads[n].reshape(-1) for n, _, _ in spec.param_spec(cfg)])


def _worker(rank, world, port, name, gb, out_dir, wire=None):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "vit-tensorflow_amd"), os.path.join(root, "tests")]
    from vit_tensorflow.parallel import GradSync, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, 1, randomize_all=True)
    rng = np.random.Generator(np.random.PCG64(3))
    img = rng.standard_normal((gb, *cfg["image_size"], 3))
    dl_global = rng.standard_normal((gb, cfg["num_classes"])) / gb       # cotangent of a GLOBAL-batch mean loss
    idx = list(shard_range(gb, rank, world))
    # each rank back-propagates the local-mean cotangent; the mean over ranks restores the global mean
    _, grads, _ = ref_torch.forward_backward(cfg, P, img[idx], dl_global[idx] * world)
    flat = torch.tensor(_flat(cfg, grads))
    table = []
    off = 0
    for n, s, _ in spec.param_spec(cfg):
        k = int(np.prod(s))
        table.append((off, k))
        off += k
    sync = GradSync(flat, bucket_elems=4096, average=True, wire_dtype=torch.bfloat16 if wire == "bf16" else None)
    sync.begin()
    for o, k in reversed(table):          # backward reports ranges from the head towards the embedding
        sync.on_ready(o, k)
    assert all(sync._launched), "every bucket must be launched once all ranges were reported"
    sync.finish()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["vit_small", "cait_small"])
def test_dp_equivalence_gloo_world2(name, tmp_path):
    world, gb = 2, 4
    port = _free_port()
    mp.spawn(_worker, args=(world, port, name, gb, str(tmp_path)), nprocs=world, join=True)
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, 1, randomize_all=True)
    rng = np.random.Generator(np.random.PCG64(3))
    img = rng.standard_normal((gb, *cfg["image_size"], 3))
    dl = rng.standard_normal((gb, cfg["num_classes"])) / gb
    _, grads, _ = ref_torch.forward_backward(cfg, P, img, dl)
    ref = _flat(cfg, grads)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1), "ranks must hold identical reduced gradients"
    assert np.abs(r0 - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


def test_dp_equivalence_gloo_world2_bf16_wire(tmp_path):
    """Gradient buckets sent as bf16 (half the xGMI bytes): both ranks end with IDENTICAL gradients, within bf16 rounding of the
    single-process gradient (each addend rounded once to 2^-9 relative, the two-rank sum once more)."""
    name, world, gb = "vit_small", 2, 4
    port = _free_port()
    mp.spawn(_worker, args=(world, port, name, gb, str(tmp_path), "bf16"), nprocs=world, join=True)
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, 1, randomize_all=True)
    rng = np.random.Generator(np.random.PCG64(3))
    img = rng.standard_normal((gb, *cfg["image_size"], 3))
    dl = rng.standard_normal((gb, cfg["num_classes"])) / gb
    _, grads, _ = ref_torch.forward_backward(cfg, P, img, dl)
    ref = _flat(cfg, grads)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1), "ranks must hold identical reduced gradients"
    assert 0 < np.abs(r0 - ref).max() <= 2.0 ** -7 * np.abs(ref).max()   # bf16 on the wire: not exact, and bounded


def test_gradsync_bucket_bookkeeping_single_process():
    from vit_tensorflow.parallel import GradSync, shard_range
    g = torch.arange(10, dtype=torch.float32)
    s = GradSync(g, bucket_elems=4)
    s.begin()
    s.on_ready(6, 4)
    assert s._launched == [False, False, True]
    s.on_ready(2, 4)
    assert s._launched == [False, True, True]
    s.on_ready(0, 2)
    assert s._launched == [True, True, True]
    s.finish()
    assert torch.equal(g, torch.arange(10, dtype=torch.float32))       # world 1: untouched
    assert list(shard_range(8, 1, 2)) == [4, 5, 6, 7]
    with pytest.raises(AssertionError):
        shard_range(7, 0, 2)
