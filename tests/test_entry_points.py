"""CPU tier: the files the driver executes directly (bench.py, __graft_entry__.py) and every helper script parse and import --
a syntax error in one of them would only show up on the GPU box otherwise."""
import glob
import importlib
import os
import py_compile
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_python_file_compiles():
    files = [os.path.join(ROOT, f) for f in ("bench.py", "__graft_entry__.py")]
    for pat in ("tools/*.py", "oracle/*.py", "oracle/tf_shim/*.py", "vit-tensorflow_amd/*.py", "vit-tensorflow_amd/vit_tensorflow/*.py", "tests/*.py"):
        files += glob.glob(os.path.join(ROOT, pat))
    assert len(files) > 40
    for f in files:
        py_compile.compile(f, doraise=True)


def test_graft_entry_exposes_build_and_smoke():
    sys.path.insert(0, ROOT)
    try:
        g = importlib.import_module("__graft_entry__")
    finally:
        sys.path.pop(0)
    assert callable(g.build) and callable(g.smoke)


def test_bench_cli_parses():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout, r.stderr[-2000:]
