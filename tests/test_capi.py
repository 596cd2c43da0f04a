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


# ---- one HIP runtime per process, whatever the load order (csrc/shim.cpp) --------------------------------
_CHILD_MAPS = '''
import ctypes as C, sys
order = sys.argv[1]
lib = None
if order == "lib_first":
    lib = C.CDLL(sys.argv[2])          # the loader alone: maps no runtime
    assert lib.pl_version() == 100     # the loader's own answer: still no runtime
    assert not [l for l in open("/proc/self/maps") if "amdhip" in l]
    import torch
else:
    import torch
    lib = C.CDLL(sys.argv[2])
built = lib.pl_hip_version_built()     # first bound call: binds the core to the runtime that is mapped NOW (and checks its major release)
assert built // 10000000 >= 6, built
mapped = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
print("MAPPED", mapped)
'''


@pytest.mark.parametrize("order", ["lib_first", "torch_first"])
def test_loader_binds_to_the_runtime_torch_brought_in_either_order(lib, order):
    """libpaule_hip.so links no HIP runtime; its first call binds libpaule_hip_core.so to the copy of libamdhip64 the process already
    has.  With torch imported -- before OR after the library was loaded -- that is torch's copy, and it stays the only one."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", _CHILD_MAPS, order, _capi.LIB_PATH], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    mapped = eval(out.stdout.split("MAPPED", 1)[1])
    assert len(mapped) == 1 and "/torch/lib/" in mapped[0], mapped


def test_core_library_links_no_hip_runtime(lib):
    import subprocess
    core = os.path.join(os.path.dirname(_capi.LIB_PATH), "libpaule_hip_core.so")
    dyn = subprocess.run(["readelf", "-d", core], capture_output=True, text=True, check=True).stdout
    assert "amdhip" not in dyn and "RUNPATH" not in dyn and "RPATH" not in dyn, dyn
    dyn = subprocess.run(["readelf", "-d", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "amdhip" not in dyn, dyn


def test_loader_reports_a_missing_core_through_the_c_abi(tmp_path):
    """A load failure is an error code + pl_last_error, never an abort: the loader alone (copied away from its core) answers every
    entry point with PL_ERR_HIP and says what it could not load."""
    import shutil
    import subprocess
    import sys
    shutil.copy(_capi.LIB_PATH, tmp_path / "libpaule_hip.so")
    code = ("import ctypes as C, sys\n"
            "lib = C.CDLL(sys.argv[1]); lib.pl_last_error.restype = C.c_char_p\n"
            "assert lib.pl_version() == 100 and lib.pl_last_error().decode() == ''   # answered by the loader, binds nothing\n"
            "rc = lib.pl_hip_version_built(); msg = lib.pl_last_error().decode()\n"
            "assert rc == 2 and 'libpaule_hip_core.so' in msg, (rc, msg)   # PL_ERR_HIP\n"
            "print('OK')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("PAULE_HIP_CORE", "PAULE_HIP_LIB")}
    out = subprocess.run([sys.executable, "-c", code, str(tmp_path / "libpaule_hip.so")], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


_CHILD_C_HOST = '''
import ctypes as C, sys
# a C host's situation: it is linked against ROCm's own runtime and has used it before this library is loaded; torch is never imported
rt = C.CDLL("/opt/rocm/lib/libamdhip64.so")
n = C.c_int(0)
assert rt.hipGetDeviceCount(C.byref(n)) == 0 and n.value >= 1, n.value
p = C.c_void_p()
assert rt.hipMalloc(C.byref(p), 1 << 20) == 0
sys.path.insert(0, sys.argv[2])
lib = C.CDLL(sys.argv[1]); lib.pl_last_error.restype = C.c_char_p
class Cfg(C.Structure):
    pass
import importlib.util
spec = importlib.util.spec_from_file_location("capi", sys.argv[2] + "/paule_amd/_capi.py"); capi = importlib.util.module_from_spec(spec); spec.loader.exec_module(capi)
cfg = capi.PlConfig(); assert lib.pl_default_config(C.byref(cfg)) == 0
cfg.batch, cfg.n_frames, cfg.pred_hidden, cfg.emb_hidden = 2, 40, 32, 32
h = C.c_void_p()
rc = lib.pl_create(C.byref(cfg), C.byref(h))
assert rc == 0, lib.pl_last_error()
assert lib.pl_destroy(h) == 0
mapped = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
assert len(mapped) == 1 and mapped[0].startswith("/opt/rocm"), mapped
print("OK", mapped)
'''


@pytest.mark.gpu
def test_c_host_with_its_own_runtime_and_library_loaded_before_torch(lib):
    """On the GPU.  (1) A host that is linked against /opt/rocm's runtime and has already used the device (no torch anywhere): the
    library binds to THAT copy, pl_create sees the device, one runtime stays mapped.  (2) The library loaded BEFORE torch in a Python
    host (r3: 'no ROCm-capable device'): the first pl_* call comes after torch's import and binds to torch's copy; a plan runs."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", _CHILD_C_HOST, _capi.LIB_PATH, ROOT], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
    code = ("import ctypes as C, sys\n"
            "sys.path.insert(0, sys.argv[2])\n"
            "lib0 = C.CDLL(sys.argv[1])          # before torch\n"
            "import torch\n"
            "from paule_amd import synthetic\n"
            "from paule_amd.engine import HipPlanner\n"
            "wl = synthetic.make_workload(2, 40, None, pred=dict(num_lstm_layers=1, hidden_size=32), emb=dict(num_lstm_layers=1, hidden_size=32))\n"
            "e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=2, n_frames=40, objective='acoustic_semvec', dtype='f32')\n"
            "e.set_targets(wl.target_mel, wl.target_semvec); e.set_cp(wl.cp0)\n"
            "l = e.step(2); e.synchronize(); assert torch.isfinite(l).all()\n"
            "mapped = sorted({x.split()[-1] for x in open('/proc/self/maps') if 'libamdhip64' in x})\n"
            "assert len(mapped) == 1 and '/torch/lib/' in mapped[0], mapped\n"
            "print('OK')\n")
    out = subprocess.run([sys.executable, "-c", code, _capi.LIB_PATH, ROOT], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
