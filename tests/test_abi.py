"""CPU tier: the C-ABI library loads, exports every symbol include/vitx.h declares, and its host-only
entry points (parameter table, validation) agree with the oracle's spec.  No compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import spec
from util import CONFIGS, oracle_cfg
from vit_tensorflow import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "vitx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(vitx_[a-z_0-9]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = N.lib()
    declared = _declared_symbols()
    assert len(declared) >= 30
    bound = {s[0] for s in N.SYMBOLS}
    for name in declared:
        assert hasattr(lib, name), f"libvitx.so does not export {name}"
        assert name in bound, f"{name} is declared in vitx.h but has no ctypes prototype"
    assert lib.vitx_version().decode().startswith("vitx")


def _cfg_struct(name):
    from util import make_engine_model
    return make_engine_model(name)._cfg


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_param_table_matches_oracle_spec(name):
    table, n = N.param_table(_cfg_struct(name))
    ps = spec.param_spec(oracle_cfg(name))
    assert [(a, tuple(s)) for a, s, _ in ps] == [(a, tuple(s)) for a, s, _ in table]
    off = 0
    for (_, s, o) in table:
        assert o == off
        off += int(np.prod(s))
    assert off == n


def test_invalid_configs_carry_the_reference_messages():
    lib = N.lib()
    cfg = _cfg_struct("vit_small")
    cfg.image_h = 65
    nt, ne = C.c_int64(), C.c_int64()
    assert lib.vitx_param_table_size(C.byref(cfg), C.byref(nt), C.byref(ne)) == N.ERR_INVALID
    assert lib.vitx_last_error().decode() == 'Image dimensions must be divisible by the patch size.'   # vit.py:136
    cfg = _cfg_struct("vit_small")
    cfg.pool = 7
    assert lib.vitx_param_table_size(C.byref(cfg), C.byref(nt), C.byref(ne)) == N.ERR_INVALID
    assert lib.vitx_last_error().decode() == 'pool type must be either cls (cls token) or mean (mean pooling)'  # vit.py:139


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = N.lib()
    h = C.c_void_p()
    cfg = _cfg_struct("vit_small")
    cfg.max_batch = 1
    rc = lib.vitx_create(C.byref(cfg), C.byref(h))
    assert rc == N.ERR_HIP and "no CPU fallback" in lib.vitx_last_error().decode()


def test_python_front_mirrors_reference_api():
    from vit_tensorflow import ViT
    from vit_tensorflow.deepvit import DeepViT
    from vit_tensorflow.cait import CaiT
    with pytest.raises(AssertionError, match='Image dimensions must be divisible by the patch size.'):
        ViT(image_size=250, patch_size=32, num_classes=10, dim=64, depth=1, heads=1, mlp_dim=64)
    with pytest.raises(AssertionError, match='pool type must be either cls'):
        ViT(image_size=256, patch_size=32, num_classes=10, dim=64, depth=1, heads=1, mlp_dim=64, pool='max')
    # README.md:45-59 constructor, verbatim kwargs
    v = ViT(image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=6, heads=16, mlp_dim=2048, dropout=0.1, emb_dropout=0.1)
    assert v.count_params() == 54622184 and v.pos_embedding.shape == (1, 65, 1024) and v.cls_token.shape == (1, 1, 1024)
    d = DeepViT(image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=6, heads=16, mlp_dim=2048, dropout=0.1, emb_dropout=0.1)
    assert any("reattn_weights" in w.name for w in d.weights)
    c = CaiT(image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=12, cls_depth=2, heads=16, mlp_dim=2048,
             dropout=0.1, emb_dropout=0.1, layer_dropout=0.05)
    assert c.pos_embedding.shape == (1, 64, 1024)                       # cait.py:168: no cls slot
    w = v.get_weights()
    w[3] = np.full_like(w[3], 0.5)
    v.set_weights(w)
    assert np.all(v.get_weights()[3] == 0.5) and np.all(v.state_dict()["patch_embedding.bias"] == 0.5)
    # initialisers (vit.py:146-147, Keras defaults, cait.py:36-43)
    sd = c.state_dict()
    assert np.all(sd["patch_transformer.0.attn.norm.gamma"] == 1) and np.all(sd["patch_transformer.0.mlp.fc1.bias"] == 0)
    assert np.allclose(sd["patch_transformer.0.attn.scale"], 0.1) and np.allclose(sd["cls_transformer.1.mlp.scale"], 0.1)
    k = sd["patch_transformer.0.mlp.fc1.kernel"]
    assert np.abs(k).max() <= np.sqrt(6.0 / (k.shape[0] + k.shape[1])) + 1e-6


# ---- environment switches: one registry (csrc/env.hip), three classes, diagnostics ignored by the release library
def test_every_environment_switch_is_registered_classified_and_documented():
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = N.debug_switches()
    names = [r[0] for r in rows]
    assert len(names) == len(set(names)) >= 40
    assert {r[1] for r in rows} == {"tuning", "path", "diag"}
    # no getenv in csrc/ outside the registry (env.hip) and the one reporting call of vitx_debug_switches (capi.hip)
    hits = []
    for f in glob.glob(os.path.join(root, "vit-tensorflow_amd", "csrc", "*")):
        for i, line in enumerate(open(f, errors="replace"), 1):
            code = line.split("//")[0]
            if re.search(r"(?<![a-z_])getenv\(", code):
                hits.append(f"{os.path.basename(f)}:{i}")
    assert sorted(h.split(":")[0] for h in hits) == ["capi.hip", "env.hip"], hits
    # every name the sources pass to vitx_env is in the table (an unregistered one aborts at run time: catch it here)
    used = set()
    for f in glob.glob(os.path.join(root, "vit-tensorflow_amd", "csrc", "*.h*")):
        used |= set(re.findall(r'(?:vitx_env|env_flag|vitx_env_flag)\("(VITX_[A-Z0-9_]+)"\)', open(f, errors="replace").read()))
    assert used <= set(names), sorted(used - set(names))
    assert set(names) <= used, f"registered but never read: {sorted(set(names) - used)}"
    # INTEGRATION.md carries the generated table
    sys_path = os.path.join(root, "tools")
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("env_table", os.path.join(sys_path, "env_table.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert mod.table() in doc, "INTEGRATION.md is stale: run `python tools/env_table.py --write`"


def test_the_release_library_ignores_diagnostic_switches():
    """A switch of class `diag` (wrong results by design) set in the environment is reported as `ignored` by the shipped library and as `set=...`
    by lib/libvitx_diag.so (the same objects with env.hip under -DVITX_DIAG); a tuning switch is honoured by both."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r]; from vit_tensorflow import _native as N; "
            "print({r[0]: r[2] for r in N.debug_switches() if r[0] in ('VITX_TN_XP', 'VITX_GEMM_AUTOTUNE', 'VITX_DV_XP')})") % os.path.join(root, "vit-tensorflow_amd")
    env = dict(os.environ, VITX_TN_XP="7", VITX_GEMM_AUTOTUNE="0")
    env.pop("VITX_LIB", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    assert eval(out) == {"VITX_TN_XP": "ignored", "VITX_GEMM_AUTOTUNE": "set=0", "VITX_DV_XP": "unset"}, out
    diag = os.path.join(root, "vit-tensorflow_amd", "lib", "libvitx_diag.so")
    if os.path.exists(diag):
        out = subprocess.run([sys.executable, "-c", code], env=dict(env, VITX_LIB=diag), capture_output=True, text=True, check=True).stdout
        assert eval(out)["VITX_TN_XP"] == "set=7", out
