"""ctypes loaders for the CPU oracle (TEST INFRASTRUCTURE).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package ``rawalign_amd`` never does.

* ``Oracle``  -- our C restatement (oracle/liboracle.so, built from source anywhere gcc is).
* ``RefDTW``  -- the reference's own ``src/dtw.cpp`` (oracle/_ref/libref_dtw.so), prebuilt in the
                 build container where /root/reference exists; ``None`` when the file is absent.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

# must match rawdtw_job_t in include/rawdtw.h (and orc_job_t / ref_job)
JOB_DTYPE = np.dtype(
    [
        ("ref_off", "<u8"),
        ("read_off", "<u4"),
        ("n", "<u4"),
        ("m", "<u4"),
        ("band_radius", "<i4"),
        ("exclude_last", "<u4"),
        ("reserved", "<u4"),
    ]
)
assert JOB_DTYPE.itemsize == 32

ANCHOR_DTYPE = np.dtype([("target_position", "<u4"), ("query_position", "<u4")])


class OrcOpt(C.Structure):
    _fields_ = [
        ("border_constraint", C.c_int),
        ("fill_method", C.c_int),
        ("band_radius_frac", C.c_float),
        ("match_bonus", C.c_float),
        ("min_score", C.c_float),
        ("fused_score", C.c_int),
    ]


class OrcStats(C.Structure):
    _fields_ = [("dtw_calls", C.c_uint64), ("cells", C.c_uint64)]


def build_oracle(march_native: bool = False) -> str:
    """(Re)build oracle/liboracle.so with gcc; returns its path."""
    args = ["make", "-C", HERE, "liboracle.so"]
    if march_native:
        args.append("ORC_MARCH=-march=native")
    subprocess.run(args, check=True, capture_output=True)
    return os.path.join(HERE, "liboracle.so")


def build_ref() -> str | None:
    """Build oracle/_ref from /root/reference (only where it exists)."""
    if not os.path.isdir("/root/reference/src"):
        return None
    subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)
    return os.path.join(HERE, "_ref", "libref_dtw.so")


def _as_f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


class Oracle:
    def __init__(self, path: str | None = None):
        path = path or os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        self.lib = L
        L.orc_dtw_global.restype = C.c_float
        L.orc_dtw_global.argtypes = [f32p, C.c_uint32, f32p, C.c_uint32, C.c_int]
        L.orc_dtw_banded.restype = C.c_float
        L.orc_dtw_banded.argtypes = [f32p, C.c_uint32, f32p, C.c_uint32, C.c_int, C.c_int]
        L.orc_dtw_banded_cellset.restype = C.c_float
        L.orc_dtw_banded_cellset.argtypes = [
            f32p, C.c_uint32, f32p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_void_p,
        ]
        L.orc_banded_cells.restype = C.c_uint64
        L.orc_banded_cells.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.orc_dtw_global_tb.restype = C.c_float
        L.orc_dtw_global_tb.argtypes = [
            f32p, C.c_uint32, f32p, C.c_uint32, C.c_int, u32p, u32p, f32p, C.POINTER(C.c_uint32),
        ]
        L.orc_dtw_directions.restype = None
        L.orc_dtw_directions.argtypes = [f32p, C.c_uint32, f32p, C.c_uint32, u8p]
        L.orc_align_chain.restype = C.c_float
        L.orc_align_chain.argtypes = [
            C.c_void_p, C.c_uint32, f32p, f32p, C.POINTER(OrcOpt), C.c_float, C.POINTER(OrcStats),
        ]
        L.orc_align_chain_cigar.restype = C.c_float
        L.orc_align_chain_cigar.argtypes = [
            C.c_void_p, C.c_uint32, f32p, f32p, C.POINTER(OrcOpt), u64p, u64p, f32p,
            C.POINTER(C.c_uint64), C.POINTER(C.c_float),
        ]
        L.orc_batch_costs.restype = None
        L.orc_batch_costs.argtypes = [C.c_void_p, C.c_uint64, f32p, f32p, f32p, C.c_int]
        L.orc_batch_costs_reps.restype = None
        L.orc_batch_costs_reps.argtypes = [C.c_void_p, C.c_uint64, f32p, f32p, f32p, C.c_int, C.c_int]

    # --- single calls -------------------------------------------------------
    def dtw_global(self, a, b, exclude_last=False) -> np.float32:
        a, b = _as_f32(a), _as_f32(b)
        return np.float32(self.lib.orc_dtw_global(a, len(a), b, len(b), int(exclude_last)))

    def dtw_banded(self, a, b, band_radius, exclude_last=False) -> np.float32:
        a, b = _as_f32(a), _as_f32(b)
        return np.float32(
            self.lib.orc_dtw_banded(a, len(a), b, len(b), int(band_radius), int(exclude_last))
        )

    def dtw_banded_cellset(self, a, b, band_radius, exclude_last=False):
        a, b = _as_f32(a), _as_f32(b)
        n_long, n_short = max(len(a), len(b)), min(len(a), len(b))
        mask = np.zeros((n_long, n_short), dtype=np.uint8)
        cells = C.c_uint64(0)
        cost = self.lib.orc_dtw_banded_cellset(
            a, len(a), b, len(b), int(band_radius), int(exclude_last), C.byref(cells),
            mask.ctypes.data_as(C.c_void_p),
        )
        return np.float32(cost), int(cells.value), mask

    def banded_cells(self, n, m, band_radius) -> int:
        return int(self.lib.orc_banded_cells(int(n), int(m), int(band_radius)))

    def dtw_global_tb(self, a, b, exclude_last=False):
        a, b = _as_f32(a), _as_f32(b)
        cap = len(a) + len(b) - 1
        pi = np.zeros(cap, np.uint32)
        pj = np.zeros(cap, np.uint32)
        pd = np.zeros(cap, np.float32)
        ln = C.c_uint32(0)
        cost = self.lib.orc_dtw_global_tb(a, len(a), b, len(b), int(exclude_last), pi, pj, pd, C.byref(ln))
        k = ln.value
        return np.float32(cost), pi[:k].copy(), pj[:k].copy(), pd[:k].copy()

    def dtw_directions(self, a, b) -> np.ndarray:
        a, b = _as_f32(a), _as_f32(b)
        d = np.zeros((len(a), len(b)), np.uint8)
        self.lib.orc_dtw_directions(a, len(a), b, len(b), d)
        return d

    # --- chains ---------------------------------------------------------------
    def align_chain(self, anchors, ref_events, read_events, opt: OrcOpt, min_score=-1e10, stats=None):
        anchors = np.ascontiguousarray(anchors, dtype=ANCHOR_DTYPE)
        st = stats if stats is not None else OrcStats()
        return np.float32(
            self.lib.orc_align_chain(
                anchors.ctypes.data_as(C.c_void_p), len(anchors), _as_f32(ref_events),
                _as_f32(read_events), C.byref(opt), C.c_float(min_score), C.byref(st),
            )
        )

    def align_chain_cigar(self, anchors, ref_events, read_events, opt: OrcOpt):
        anchors = np.ascontiguousarray(anchors, dtype=ANCHOR_DTYPE)
        cap = 0
        parts = len(anchors) - 1
        if opt.border_constraint == 0:
            cap = int(anchors[0]["query_position"] - anchors[-1]["query_position"] + 1) + int(
                anchors[0]["target_position"] - anchors[-1]["target_position"] + 1
            )
        else:
            for p in range(parts):
                s, e = anchors[parts - p], anchors[parts - p - 1]
                cap += int(e["query_position"] - s["query_position"] + 1) + int(
                    e["target_position"] - s["target_position"] + 1
                )
        cap = max(cap, 1)
        pi = np.zeros(cap, np.uint64)
        pj = np.zeros(cap, np.uint64)
        pd = np.zeros(cap, np.float32)
        ln = C.c_uint64(0)
        cost = C.c_float(0)
        score = self.lib.orc_align_chain_cigar(
            anchors.ctypes.data_as(C.c_void_p), len(anchors), _as_f32(ref_events), _as_f32(read_events),
            C.byref(opt), pi, pj, pd, C.byref(ln), C.byref(cost),
        )
        if ln.value == 2**64 - 1:
            raise AssertionError("global+banded+cigar is not implemented (rmap.cpp:223-225)")
        k = ln.value
        return np.float32(score), np.float32(cost.value), pi[:k].copy(), pj[:k].copy(), pd[:k].copy()

    def batch_costs(self, jobs, events, ref, nthreads=1, reps=1) -> np.ndarray:
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        out = np.zeros(len(jobs), np.float32)
        self.lib.orc_batch_costs_reps(
            jobs.ctypes.data_as(C.c_void_p), len(jobs), _as_f32(events), _as_f32(ref), out, int(nthreads), int(reps)
        )
        return out


class RefDTW:
    """The reference's own compiled dtw.cpp (oracle/_ref/libref_dtw.so)."""

    def __init__(self, path: str | None = None):
        path = path or os.path.join(HERE, "_ref", "libref_dtw.so")
        L = C.CDLL(path)
        self.lib = L
        for name in ("ref_dtw_global", "ref_dtw_global_slow"):
            fn = getattr(L, name)
            fn.restype = C.c_float
            fn.argtypes = [f32p, C.c_uint32, f32p, C.c_uint32, C.c_int]
        for name in ("ref_dtw_banded", "ref_dtw_slantedbanded"):
            fn = getattr(L, name)
            fn.restype = C.c_float
            fn.argtypes = [f32p, C.c_uint32, f32p, C.c_uint32, C.c_int, C.c_int]
        L.ref_dtw_global_tb.restype = C.c_float
        L.ref_dtw_global_tb.argtypes = [
            f32p, C.c_uint32, f32p, C.c_uint32, C.c_int, u32p, u32p, f32p, C.POINTER(C.c_uint32),
        ]
        L.ref_batch_costs.restype = None
        L.ref_batch_costs.argtypes = [C.c_void_p, C.c_uint64, f32p, f32p, f32p, C.c_int]
        L.ref_batch_costs_reps.restype = None
        L.ref_batch_costs_reps.argtypes = [C.c_void_p, C.c_uint64, f32p, f32p, f32p, C.c_int, C.c_int]

    @staticmethod
    def available(path: str | None = None) -> bool:
        return os.path.exists(path or os.path.join(HERE, "_ref", "libref_dtw.so"))

    def dtw_global(self, a, b, exclude_last=False):
        a, b = _as_f32(a), _as_f32(b)
        return np.float32(self.lib.ref_dtw_global(a, len(a), b, len(b), int(exclude_last)))

    def dtw_global_slow(self, a, b, exclude_last=False):
        a, b = _as_f32(a), _as_f32(b)
        return np.float32(self.lib.ref_dtw_global_slow(a, len(a), b, len(b), int(exclude_last)))

    def dtw_banded(self, a, b, band_radius, exclude_last=False):
        a, b = _as_f32(a), _as_f32(b)
        return np.float32(self.lib.ref_dtw_banded(a, len(a), b, len(b), int(band_radius), int(exclude_last)))

    def dtw_slantedbanded(self, a, b, band_radius, exclude_last=False):
        a, b = _as_f32(a), _as_f32(b)
        return np.float32(
            self.lib.ref_dtw_slantedbanded(a, len(a), b, len(b), int(band_radius), int(exclude_last))
        )

    def dtw_global_tb(self, a, b, exclude_last=False):
        a, b = _as_f32(a), _as_f32(b)
        cap = len(a) + len(b) - 1
        pi = np.zeros(cap, np.uint32)
        pj = np.zeros(cap, np.uint32)
        pd = np.zeros(cap, np.float32)
        ln = C.c_uint32(0)
        cost = self.lib.ref_dtw_global_tb(a, len(a), b, len(b), int(exclude_last), pi, pj, pd, C.byref(ln))
        k = ln.value
        return np.float32(cost), pi[:k].copy(), pj[:k].copy(), pd[:k].copy()

    def batch_costs(self, jobs, events, ref, nthreads=1, reps=1) -> np.ndarray:
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        out = np.zeros(len(jobs), np.float32)
        self.lib.ref_batch_costs_reps(
            jobs.ctypes.data_as(C.c_void_p), len(jobs), _as_f32(events), _as_f32(ref), out, int(nthreads), int(reps)
        )
        return out
