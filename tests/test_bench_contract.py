"""bench.py's contract with the driver: the FLOP model behind `value` / `roofline` (CPU tier) and the one-line JSON schema of a
real run (GPU tier, small batch so that it finishes in seconds)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("vitx_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_flop_model_matches_the_survey_figures():
    b = _bench_module()
    # ViT-B/16 224 = 105.383 GFLOP fwd+bwd per image, ViT-L/16 224 = 369.328, README cfg1 = 21.155
    assert abs(b.flops_per_image(b.WORKLOADS["vit_b16_224"]) / 1e9 - 105.383) < 5e-3
    assert abs(b.flops_per_image(b.WORKLOADS["vit_l16_224"]) / 1e9 - 369.328) < 5e-3
    assert abs(b.flops_per_image(b.WORKLOADS["vit_readme_256"]) / 1e9 - 21.155) < 5e-3
    assert b.MFMA_BF16_PEAK == 2516.6e12 and b.HBM_PEAK == 8.0e12
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert "images/sec" in base["metric"]


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "images/sec" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2516.6 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["native", "torch", "native-unavailable"])
def test_bench_data_parallel_path_on_a_group_of_one(impl):
    """The N > 1 code path of bench.py (rendezvous, weight broadcast, the gradient exchange overlapped with the backward pass, the join) on the one GPU
    a test box has: VITX_FORCE_DP=1 forms an RCCL group of one.  native = the library's own bucketed exchange (csrc/comm.hip), torch = GradSync."""
    env = dict(os.environ, VITX_FORCE_DP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    if impl == "native-unavailable":   # the native exchange reports an error at start-up: every rank falls back to the torch exchange (own NCCL group)
        env["VITX_BENCH_SIMULATE_NATIVE_FAILURE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-profile",
                          "--bucket-mb", "16", "--dp-impl", "torch" if impl == "torch" else "native"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    if impl == "native-unavailable":
        assert "unavailable" in d["config"]["dp_impl"] and "exchange" not in d["config"]
    if impl == "native":
        ex = d["config"]["exchange"]
        assert ex["buckets"] >= 2 and 1 <= ex["sent_during_backward"] <= ex["buckets"], ex   # 346 MB of gradients in 16-MiB buckets, most sent early
