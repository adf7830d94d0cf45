"""rawalign_amd -- MI355X-native DTW alignment engine behind RawAlign's align_chain.

The package holds only what the hot path needs: ``csrc/`` (hand-written gfx950 kernels and the
C ABI of ``include/rawdtw.h``), a ctypes binding to that ABI, and the host-side mirror of the
reference interface for this path (``dtw.hpp`` function names, ``align_chain`` semantics).

There is no CPU fallback: importing works anywhere, but every compute call needs
``librawdtw.so`` and a HIP device and raises otherwise.
"""
from ._lib import LibraryMissing, RawDTWError, load_library, library_path  # noqa: F401
from .dtw import ANCHOR_DTYPE, JOB_DTYPE, RAWDTW_FULL, DtwResult, Engine, Plan  # noqa: F401
from .align import (  # noqa: F401
    RI_M_DTW_BORDER_CONSTRAINT_GLOBAL,
    RI_M_DTW_BORDER_CONSTRAINT_SPARSE,
    RI_M_DTW_FILL_METHOD_BANDED,
    RI_M_DTW_FILL_METHOD_FULL,
    Batch,
    CandidateBatch,
    Chain,
    MapOpt,
    ReadCandidates,
    align_chain,
    evaluate_reads,
)

__all__ = [
    "Engine", "Plan", "DtwResult", "JOB_DTYPE", "ANCHOR_DTYPE", "RAWDTW_FULL",
    "MapOpt", "Chain", "Batch", "CandidateBatch", "ReadCandidates", "align_chain", "evaluate_reads",
    "load_library", "library_path", "LibraryMissing", "RawDTWError",
]

DEFAULT_FOLD_MODE = 4  # rawdtw_set_option("fold_mode"): the library's default chain-fold kernel
