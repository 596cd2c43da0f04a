"""ctypes binding of libpaule_hip.so (C-ABI declared in include/paule_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded the
import of the engine fails loudly (``HipLibraryError``).  The error convention mirrors the
reference's one C library: a nonzero return code becomes a ``ValueError``
(paule/util.py:33-34, :235-236, :246-247).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpaule_hip.so")

PL_F32, PL_BF16 = 0, 1
PL_OBJ = {"acoustic": 0, "acoustic_semvec": 1, "semvec": 2}
PL_MODEL_PRED, PL_MODEL_EMBED, PL_MODEL_INVERSE = 0, 1, 2
PL_MODEL_CP_TUBE, PL_MODEL_TUBE_MEL, PL_MODEL_TUBE_EMBED = 3, 4, 5
PL_CONV_MEL, PL_CONV_RES, PL_CONV_RW = 0, 1, 2
PL_LOSS_COLS = 8

# every symbol include/paule_hip.h declares
EXPORTED_SYMBOLS = (
    "pl_default_config", "pl_create", "pl_destroy", "pl_set_lstm_weights", "pl_set_linear",
    "pl_set_speech_classifier", "pl_set_targets", "pl_set_cp", "pl_set_past_cp", "pl_reset_optimizer", "pl_step", "pl_synchronize", "pl_get_cp",
    "pl_get_pred", "pl_get_pred_frames", "pl_get_tube_pred", "pl_embed_tube", "pl_embed_mel", "pl_debug_read", "pl_bench_kernel", "pl_plan_info", "pl_device_bytes", "pl_flops_per_iteration",
    "pl_train_pred_step", "pl_reset_pred_optimizer", "pl_get_lstm_weights", "pl_get_linear",
    "pl_set_inverse_conv", "pl_inverse_forward", "pl_set_embedder_output", "pl_set_embedder_conv",
    "pl_get_pred_optimizer_state", "pl_set_pred_optimizer_state", "pl_get_pred_optimizer_step", "pl_set_pred_optimizer_step",
    "pl_train_model_step", "pl_reset_model_optimizer", "pl_get_model_optimizer_state", "pl_set_model_optimizer_state",
    "pl_get_model_optimizer_step", "pl_set_model_optimizer_step",
    "pl_last_error", "pl_version", "pl_hip_version_built",
)


class HipLibraryError(RuntimeError):
    """libpaule_hip.so is missing / unloadable: the product has no other execution path."""


class PlConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("batch", C.c_int32), ("n_frames", C.c_int32), ("cp_dim", C.c_int32),
        ("mel_dim", C.c_int32), ("sem_dim", C.c_int32), ("pred_layers", C.c_int32), ("pred_hidden", C.c_int32),
        ("emb_layers", C.c_int32), ("emb_hidden", C.c_int32), ("dtype", C.c_int32), ("objective", C.c_int32),
        ("w_mel", C.c_float), ("w_sem", C.c_float), ("w_vel", C.c_float), ("w_jerk", C.c_float), ("w_ll", C.c_float),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("clamp_lo", C.c_float), ("clamp_hi", C.c_float), ("smiling", C.c_int32), ("device", C.c_int32),
        ("use_graph", C.c_int32), ("stream", C.c_void_p),
        ("inv_layers", C.c_int32), ("inv_hidden", C.c_int32), ("inv_mel_blocks", C.c_int32), ("inv_res_blocks", C.c_int32),
        ("emb_post_size", C.c_int32), ("emb_mel_blocks", C.c_int32),
        ("tube_dim", C.c_int32), ("cp_tube_layers", C.c_int32), ("cp_tube_hidden", C.c_int32), ("tube_mel_layers", C.c_int32),
        ("tube_mel_hidden", C.c_int32), ("tube_emb_layers", C.c_int32), ("tube_emb_hidden", C.c_int32),
    ]


_lib = None


def load_library(path: str | None = None):
    """Loads libpaule_hip.so once and declares the prototypes.  Raises HipLibraryError if absent.

    libpaule_hip.so is a LOADER (csrc/shim.cpp): it links no HIP runtime.  On the first pl_* call that needs the kernels it binds
    libpaule_hip_core.so to the HIP runtime the process already has (PyTorch's, if torch was imported -- in whatever order) or, when there
    is none, to /opt/rocm's.  So nothing here depends on import order any more; loading the library, pl_version() and
    pl_last_error() touch no runtime at all (the loader answers them itself); every OTHER entry point -- pl_default_config included -- binds,
    and the loader refuses a runtime of another major release than the kernels were compiled for (pl_hip_version_built).  PAULE_HIP_LIB = another build of the CORE (diagnostic builds: libpaule_hip_stamps.so, an A/B
    build) or another directory's libpaule_hip.so."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("PAULE_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise HipLibraryError(
            f"{p} not found: build it with `make -C paule_amd/csrc` (or `python -c 'import __graft_entry__ as g; "
            "g.build()'`).  paule_amd has no CPU fallback.")
    if os.path.basename(p) != "libpaule_hip.so":   # a core variant: the shipped loader with that core
        os.environ["PAULE_HIP_CORE"] = os.path.abspath(p)
        p = LIB_PATH
    core = os.environ.get("PAULE_HIP_CORE", os.path.join(os.path.dirname(os.path.abspath(p)), "libpaule_hip_core.so"))
    if not os.path.exists(p) or not os.path.exists(core):
        raise HipLibraryError(f"{p if not os.path.exists(p) else core} not found: build it with `make -C paule_amd/csrc`.  paule_amd has no CPU fallback.")
    try:
        lib = C.CDLL(p)
    except OSError as e:
        raise HipLibraryError(f"cannot load {p}: {e}") from e

    vp, fp, ip = C.c_void_p, C.c_void_p, C.c_void_p  # device pointers travel as raw addresses
    lib.pl_last_error.restype = C.c_char_p
    lib.pl_last_error.argtypes = []
    lib.pl_version.restype = C.c_int
    lib.pl_version.argtypes = []
    lib.pl_default_config.argtypes = [C.POINTER(PlConfig)]
    lib.pl_create.argtypes = [C.POINTER(PlConfig), C.POINTER(vp)]
    lib.pl_destroy.argtypes = [vp]
    lib.pl_set_lstm_weights.argtypes = [vp, C.c_int, C.c_int, fp, fp, fp, fp]
    lib.pl_set_linear.argtypes = [vp, C.c_int, fp, fp]
    lib.pl_set_speech_classifier.argtypes = [vp, fp, fp, C.c_float]
    lib.pl_set_targets.argtypes = [vp, fp, fp]
    lib.pl_set_cp.argtypes = [vp, fp]
    lib.pl_set_past_cp.argtypes = [vp, fp, C.c_int, C.c_int]
    lib.pl_reset_optimizer.argtypes = [vp]
    lib.pl_step.argtypes = [vp, C.c_int, fp, fp]
    lib.pl_synchronize.argtypes = [vp]
    lib.pl_get_cp.argtypes = [vp, fp]
    lib.pl_get_pred.argtypes = [vp, fp, fp]
    lib.pl_get_pred_frames.argtypes = [vp, fp]
    lib.pl_get_tube_pred.argtypes = [vp, fp, fp, fp]
    lib.pl_embed_tube.argtypes = [vp, fp, fp, fp]
    lib.pl_embed_mel.argtypes = [vp, fp, ip, fp]
    lib.pl_debug_read.argtypes = [vp, C.c_char_p, fp, C.c_int64, C.POINTER(C.c_int64)]
    lib.pl_bench_kernel.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
    lib.pl_train_pred_step.argtypes = [vp, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, fp]
    lib.pl_reset_pred_optimizer.argtypes = [vp]
    lib.pl_get_lstm_weights.argtypes = [vp, C.c_int, C.c_int, fp, fp, fp, fp]
    lib.pl_get_linear.argtypes = [vp, C.c_int, fp, fp]
    lib.pl_set_inverse_conv.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp, fp]
    lib.pl_inverse_forward.argtypes = [vp, fp, C.c_int, fp, C.c_int]
    lib.pl_set_embedder_output.argtypes = [vp, fp, fp]
    lib.pl_set_embedder_conv.argtypes = [vp, C.c_int, C.c_int, fp, fp]
    lib.pl_get_pred_optimizer_state.argtypes = [vp, C.c_int, C.c_int, fp, fp, fp, fp]
    lib.pl_set_pred_optimizer_state.argtypes = [vp, C.c_int, C.c_int, fp, fp, fp, fp]
    lib.pl_get_pred_optimizer_step.restype = C.c_int64
    lib.pl_get_pred_optimizer_step.argtypes = [vp]
    lib.pl_set_pred_optimizer_step.argtypes = [vp, C.c_int64]
    lib.pl_train_model_step.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, fp]
    lib.pl_reset_model_optimizer.argtypes = [vp, C.c_int]
    lib.pl_get_model_optimizer_state.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp]
    lib.pl_set_model_optimizer_state.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp]
    lib.pl_get_model_optimizer_step.restype = C.c_int64
    lib.pl_get_model_optimizer_step.argtypes = [vp, C.c_int]
    lib.pl_set_model_optimizer_step.argtypes = [vp, C.c_int, C.c_int64]
    lib.pl_plan_info.argtypes = [vp, C.POINTER(C.c_int32), C.c_int]
    lib.pl_plan_info.restype = C.c_int
    lib.pl_device_bytes.restype = C.c_int64
    lib.pl_device_bytes.argtypes = [vp]
    lib.pl_flops_per_iteration.restype = C.c_double
    lib.pl_flops_per_iteration.argtypes = [vp]
    for name in ("pl_default_config", "pl_create", "pl_destroy", "pl_set_lstm_weights", "pl_set_linear", "pl_set_speech_classifier",
                 "pl_set_targets", "pl_set_cp", "pl_set_past_cp", "pl_reset_optimizer", "pl_step", "pl_get_cp",
                 "pl_get_pred", "pl_get_pred_frames", "pl_get_tube_pred", "pl_embed_tube", "pl_embed_mel", "pl_debug_read", "pl_bench_kernel", "pl_synchronize", "pl_train_pred_step",
                 "pl_reset_pred_optimizer", "pl_get_lstm_weights", "pl_get_linear", "pl_set_inverse_conv", "pl_inverse_forward", "pl_set_embedder_output", "pl_set_embedder_conv",
                 "pl_get_pred_optimizer_state", "pl_set_pred_optimizer_state", "pl_set_pred_optimizer_step",
                 "pl_train_model_step", "pl_reset_model_optimizer", "pl_get_model_optimizer_state", "pl_set_model_optimizer_state",
                 "pl_set_model_optimizer_step"):
        getattr(lib, name).restype = C.c_int
    if path is None:
        _lib = lib
    return lib


def check(lib, rc: int, what: str = ""):
    """nonzero return code -> ValueError with the library's message (paule/util.py:33-34 convention)."""
    if rc != 0:
        msg = lib.pl_last_error()
        raise ValueError(f"{what}: {msg.decode() if msg else 'error'} (code {rc})")
