"""paule_amd -- MI355X-native (gfx950) gradient-planning path of PAULE.

Drop-in for ``paule.Paule.plan_resynth()`` / ``paule.models.ForwardModel`` /
``paule.models.EmbeddingModel`` / ``paule.models.InverseModelMelTimeSmoothResidual`` on that path only: hand-written HIP kernels behind a C-ABI
(``include/paule_hip.h``, ``paule_amd/csrc``).  No CPU fallback: the engine raises
``HipLibraryError`` when libpaule_hip.so or the GPU is missing.
"""
from ._capi import HipLibraryError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):   # lazy: importing the package must not need torch.cuda
    if name in ("Paule", "PlanningResults", "PlanningResultsWithSpeechClassifier", "PlanningResultsWithSomatosensory"):
        from . import paule as _p
        return getattr(_p, name)
    if name == "HipPlanner":
        from .engine import HipPlanner
        return HipPlanner
    if name in ("ForwardModel", "EmbeddingModel", "InverseModelMelTimeSmoothResidual", "MelEmbeddingModelMelSmoothResidualUpsampling"):
        from . import models as _m
        return getattr(_m, name)
    raise AttributeError(name)
