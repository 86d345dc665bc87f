import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vit-tensorflow_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library must exist for both tiers (CPU tier: load + symbol + host-only calls)."""
    lib = os.path.join(PKG, "lib", "libvitx.so")
    if not os.path.exists(lib):
        sys.path.insert(0, PKG)
        import build as _b   # vit-tensorflow_amd/build.py (hipcc cross-compiles without a GPU)
        _b.build()
    yield


@pytest.fixture(autouse=True)
def _print_observed_gate_errors():
    """After each test: the worst error each bf16-sized gate saw (tests/util.py:gate)."""
    yield
    try:
        from util import report_gates
        report_gates()
    except Exception:
        pass
