"""SURVEY.md section 8(b), threading: one handle is driven by one host thread at a time, DIFFERENT handles may be driven from different threads
concurrently (no global mutable state besides the per-thread last-error string).  Five handles (ViT, DeepViT, CaiT and two models at real widths, bf16 mode -- the path with the
process-wide caches: per-shape GEMM variant table, GELU table, per-kernel LDS attributes) step concurrently from five Python threads (ctypes
releases the GIL for the duration of every C call); every step must give the bits of the same sequence run alone."""
import threading

import numpy as np
import pytest

from oracle import spec
from util import make_engine_model, oracle_cfg, rand_images

pytestmark = pytest.mark.gpu

NAMES = ["vit_bf16_small", "deepvit_bf16_small", "cait_bf16_small", "cfg2_vit_b16", "cfg1_readme"]   # the last two: widths at which the per-shape GEMM measurement runs
STEPS = 12


def _sequence(name, out, errors=None):
    try:
        cfg = oracle_cfg(name)
        P = spec.init_params(cfg, 1, randomize_all=True)
        m = make_engine_model(name, "bf16", 4, P)
        res = []
        for s in range(STEPS):
            b = 1 + (s * 7 + len(name)) % 4                      # the batch changes from step to step
            img = rand_images(cfg, b, seed=100 + s)
            dl = (np.random.default_rng(s).standard_normal((b, cfg["num_classes"])) / b).astype(np.float32)
            logits = m(img, training=False)
            grads, _ = m.backward(dl)
            res.append((logits.copy(), {k: v.copy() for k, v in grads.items()}))
        out[name] = res
    except Exception as ex:   # noqa: BLE001 -- reported by the main thread
        if errors is not None:
            errors.append((name, repr(ex)))
        else:
            raise


def test_handles_stepped_from_concurrent_threads_give_the_single_thread_bits():
    alone = {}
    for n in NAMES:
        _sequence(n, alone)
    for rep in range(3):
        together, errors = {}, []
        threads = [threading.Thread(target=_sequence, args=(n, together, errors)) for n in NAMES]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for n in NAMES:
            for s, ((l0, g0), (l1, g1)) in enumerate(zip(alone[n], together[n])):
                assert np.array_equal(l0, l1), (rep, n, s, "logits")
                for k in g0:
                    assert np.array_equal(g0[k], g1[k]), (rep, n, s, k)
