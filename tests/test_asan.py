"""The C ABI under AddressSanitizer (host side: argument validation, parameter tables, engine bookkeeping, the ctypes marshalling of
vit_tensorflow/_native.py).  `vit-tensorflow_amd/build.py --asan` builds libvitx_asan.so (cached; ~2 min the first time); the tests of
tests/test_abi.py re-run in a subprocess with the sanitizer runtime preloaded.  (CPU only: with a GPU present ROCm's sanitizer runtime intercepts
hsa_amd_memory_pool_allocate for its device-side mode and aborts HIP's start-up on a node that is not set up for it -- xnack+, sanitized ROCm
libraries -- so the end-to-end run under the sanitizer is not part of the GPU tier; measured once, profiles/r3/asan_with_gpu_r3.log.)"""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_py():
    spec = importlib.util.spec_from_file_location("vitx_build", os.path.join(ROOT, "vit-tensorflow_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run_under_asan(pytest_args):
    b = _build_py()
    if not os.path.exists(b.ASAN_RT):
        pytest.skip("no AddressSanitizer runtime in this toolchain")
    lib = b.build_asan()
    env = dict(os.environ, LD_PRELOAD=b.ASAN_RT, VITX_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=0:protect_shadow_gap=0:halt_on_error=1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", *pytest_args], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=1500)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]
    return out


def test_c_abi_host_paths_are_clean_under_address_sanitizer():
    out = _run_under_asan(["tests/test_abi.py"])
    assert " passed" in out
