"""MI355X-native cut scoring / selection for low-dimensional PSD cuts.

Hot path of rb2309/SDPCutSel-via-NN (cut_select_qp.py:543-797, neural_nets/NNs.so) as
hand-written gfx950 HIP kernels behind a C-ABI (include/sdpcut.h), with a host mirror of
the reference's ``CutSolver`` selection methods.
"""
from ._capi import EIG, NN, Scorer, SdpCutError, load_library  # noqa: F401
from .cut_solver import (CutSolver, CutSolverQCQP, GpuCutSelectionMixin, RankList,  # noqa: F401
                         make_dropin_classes)

__all__ = ["Scorer", "SdpCutError", "load_library", "EIG", "NN", "CutSolver", "CutSolverQCQP",
           "GpuCutSelectionMixin", "RankList", "make_dropin_classes"]
