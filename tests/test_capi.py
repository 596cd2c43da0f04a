"""C-ABI library: loads, exports every symbol include/paule_hip.h declares, and fails loudly without a GPU.
No compute calls here (CPU suite)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from paule_amd import _capi


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "paule_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pl_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _capi.load_library()


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_capi.EXPORTED_SYMBOLS)


def test_exports_every_declared_symbol(lib):
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_version_and_default_config(lib):
    assert lib.pl_version() == 100
    cfg = _capi.PlConfig()
    assert lib.pl_default_config(C.byref(cfg)) == 0
    assert cfg.struct_size == C.sizeof(_capi.PlConfig)
    # reference constants: paule/paule.py:592-597 (weights), :391 (lr), :1202 (clamp); Paule default models :124, :167
    assert (cfg.w_mel, cfg.w_sem, cfg.w_vel, cfg.w_jerk, cfg.w_ll) == (5.0, 10.0, 80.0, 400.0, 100000.0)
    assert abs(cfg.lr - 0.01) < 1e-9 and abs(cfg.clamp_lo + 1.05) < 1e-6 and abs(cfg.clamp_hi - 1.05) < 1e-6
    assert (cfg.cp_dim, cfg.mel_dim, cfg.sem_dim) == (30, 60, 300)
    assert (cfg.pred_layers, cfg.pred_hidden, cfg.emb_layers, cfg.emb_hidden) == (1, 720, 2, 720)


def test_config_validation_errors(lib):
    """nonzero code + message, never abort (error convention of paule/util.py:33-34)."""
    cfg = _capi.PlConfig()
    lib.pl_default_config(C.byref(cfg))
    h = C.c_void_p()
    cfg.struct_size = 4
    assert lib.pl_create(C.byref(cfg), C.byref(h)) != 0
    assert b"struct_size" in lib.pl_last_error()
    lib.pl_default_config(C.byref(cfg))
    cfg.n_frames = 8
    assert lib.pl_create(C.byref(cfg), C.byref(h)) != 0
    assert b"n_frames" in lib.pl_last_error()
    lib.pl_default_config(C.byref(cfg))
    cfg.n_frames, cfg.objective = 40, 7
    assert lib.pl_create(C.byref(cfg), C.byref(h)) != 0
    assert b"objective has to be one of" in lib.pl_last_error()
    with pytest.raises(ValueError):
        _capi.check(lib, lib.pl_create(C.byref(cfg), C.byref(h)), "pl_create")
    assert not h.value


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_capi.HipLibraryError):
        _capi.load_library(str(tmp_path / "libpaule_hip.so"))


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from paule_amd import models, synthetic
    from paule_amd.engine import HipPlanner
    pred_sd, emb_sd = synthetic.make_models(None, pred=dict(num_lstm_layers=1, hidden_size=8),
                                            emb=dict(num_lstm_layers=1, hidden_size=8))
    with pytest.raises(_capi.HipLibraryError):
        HipPlanner(pred_sd, emb_sd, batch=1, n_frames=20)
    with pytest.raises(_capi.HipLibraryError):
        models.ForwardModel(hidden_size=8, num_lstm_layers=1)(torch.zeros(1, 20, 30))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "paule_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
