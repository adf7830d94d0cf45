"""Engine: Python face of the C ABI, with the reference's dtw.hpp function names
(src/dtw.hpp:21-29) as methods so that parity tests read like src/check_dtw.cpp."""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass

import numpy as np

from ._lib import PlanInfo, RawDTWError, load_library

RAWDTW_FULL = -1

# rawdtw_job_t (include/rawdtw.h), 32 bytes
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
# ri_anchor_t (src/rmap.h:21-27)
ANCHOR_DTYPE = np.dtype([("target_position", "<u4"), ("query_position", "<u4")])


@dataclass
class DtwResult:
    """dtw_result (src/dtw.hpp:16-19): cost + alignment as (i, j, difference) columns."""

    cost: np.float32
    i: np.ndarray
    j: np.ndarray
    difference: np.ndarray

    def __len__(self):
        return len(self.i)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _f32(x) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.float32)


def plan_dry_run(jobs: np.ndarray, n_events: int, n_reference: int, threads: int = 0, options=None):
    """Host half of plan creation only (include/rawdtw.h: rawdtw_plan_dry_run): bins `jobs`, checks the plan's
    invariants and returns (info dict, n_tiles).  Touches no device and scores nothing."""
    lib = load_library()
    jobs = np.ascontiguousarray(jobs, JOB_DTYPE)
    options = dict(options or {})
    names = (C.c_char_p * max(len(options), 1))(*[k.encode() for k in options])
    vals = (C.c_int64 * max(len(options), 1))(*[int(v) for v in options.values()])
    info = PlanInfo()
    n_tiles = C.c_uint64()
    msg = C.create_string_buffer(512)
    st = lib.rawdtw_plan_dry_run(int(n_events), int(n_reference), _ptr(jobs), len(jobs), int(threads),
                                 C.cast(names, C.c_void_p), C.cast(vals, C.c_void_p), len(options), C.byref(info),
                                 C.byref(n_tiles), C.cast(msg, C.c_void_p), 512)
    if st != 0:
        raise RawDTWError(st, msg.value.decode())
    return {f: getattr(info, f) for f, _ in PlanInfo._fields_}, int(n_tiles.value)


class Plan:
    """A size-binned batch resident on the device (rawdtw_plan)."""

    def __init__(self, engine: "Engine", handle, n_jobs: int):
        self.engine = engine
        self._h = handle
        self.n_jobs = n_jobs
        engine._children.add(self)

    def info(self) -> dict:
        pi = PlanInfo()
        self.engine._check(self.engine.lib.rawdtw_plan_info(self._h, C.byref(pi)))
        return {k: int(getattr(pi, k)) for k, _ in PlanInfo._fields_}

    def run(self):
        self.engine._check(self.engine.lib.rawdtw_plan_run(self.engine._ctx, self._h))

    def run_timed(self):
        """Run once with a HIP event around every launch (on the engine's stream).
        Returns [(kind, param, ms), ...]."""
        n = self.info()["n_launches"]
        ms = np.zeros(max(n, 1), np.float32)
        kind = np.zeros(max(n, 1), np.uint32)
        self.engine._check(
            self.engine.lib.rawdtw_plan_run_timed(self.engine._ctx, self._h, _ptr(ms), _ptr(kind), n)
        )
        return [(int(kind[k] & 0xFF), int(kind[k] >> 8), float(ms[k])) for k in range(n)]

    def fetch(self) -> np.ndarray:
        out = np.empty(self.n_jobs, np.float32)
        self.engine._check(self.engine.lib.rawdtw_plan_fetch(self.engine._ctx, self._h, _ptr(out)))
        return out

    def close(self):
        if self._h is not None:
            self.engine.lib.rawdtw_plan_destroy(self._h)  # (safe in any order: rawdtw_destroy detaches live plans)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One rawdtw_ctx (one HIP device, one stream)."""

    KIND_NAMES = {1: "band_lane", 2: "band_wave_lds", 3: "full_wave", 4: "full_tb", 5: "tb_walk", 6: "chain_fold",
                  7: "read_select", 8: "band_wreg", 9: "band_lane_hi", 10: "band_merged"}

    def __init__(self, device: int = 0):
        self.lib = load_library()
        ctx = C.c_void_p()
        st = self.lib.rawdtw_create(int(device), C.byref(ctx))
        if st != 0:
            raise RawDTWError(st, self.lib.rawdtw_status_string(st).decode())
        self._ctx = ctx
        self.device = device
        self._keep = []  # arrays / tensors the context points at
        # plans and batches of this context: closed with it, so that their device memory goes back at a known point (the
        # C ABI itself tolerates any order: rawdtw_destroy detaches what is still alive)
        self._children = weakref.WeakSet()

    # -- plumbing ---------------------------------------------------------------
    def _check(self, st: int):
        if st != 0:
            raise RawDTWError(st, self.lib.rawdtw_last_error(self._ctx).decode() or
                              self.lib.rawdtw_status_string(st).decode())

    def close(self):
        if getattr(self, "_ctx", None) is not None:
            for child in list(self._children):
                child.close()
            self.lib.rawdtw_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self.lib.rawdtw_sync(self._ctx))

    def set_option(self, name: str, value: int):
        self._check(self.lib.rawdtw_set_option(self._ctx, name.encode(), int(value)))

    def stream_handle(self) -> int:
        s = C.c_void_p()
        self._check(self.lib.rawdtw_stream(self._ctx, C.byref(s)))
        return s.value or 0

    # -- arenas -----------------------------------------------------------------
    def upload_reference(self, forward_signals, reverse_signals):
        """ri_idx_t.forward_signals / reverse_signals (src/rawindex.h:32-34), one array per sequence."""
        fwd = [_f32(x) for x in forward_signals]
        rev = [_f32(x) for x in reverse_signals]
        assert len(fwd) == len(rev) and all(len(f) == len(r) for f, r in zip(fwd, rev))
        n = len(fwd)
        fp = (C.c_void_p * n)(*[x.ctypes.data for x in fwd])
        rp = (C.c_void_p * n)(*[x.ctypes.data for x in rev])
        ln = np.array([len(x) for x in fwd], np.uint32)
        self._check(self.lib.rawdtw_upload_reference(self._ctx, n, fp, rp, _ptr(ln)))

    def reference_offset(self, seq: int, strand: int) -> int:
        off = C.c_uint64()
        self._check(self.lib.rawdtw_reference_offset(self._ctx, seq, strand, C.byref(off)))
        return off.value

    def set_reference_device(self, data_ptr: int, n_floats: int, keepalive=None):
        self._check(self.lib.rawdtw_set_reference_device(self._ctx, C.c_void_p(data_ptr), n_floats))
        self._keep_ref = keepalive

    def upload_events(self, events):
        ev = _f32(events)
        self._check(self.lib.rawdtw_upload_events(self._ctx, _ptr(ev), len(ev)))

    def set_events_device(self, data_ptr: int, n_floats: int, keepalive=None):
        self._check(self.lib.rawdtw_set_events_device(self._ctx, C.c_void_p(data_ptr), n_floats))
        self._keep_ev = keepalive

    # -- batches ----------------------------------------------------------------
    def plan(self, jobs) -> Plan:
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        h = C.c_void_p()
        self._check(self.lib.rawdtw_plan_create(self._ctx, _ptr(jobs), len(jobs), C.byref(h)))
        return Plan(self, h, len(jobs))

    def score_batch(self, jobs, events) -> np.ndarray:
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        ev = _f32(events)
        out = np.empty(len(jobs), np.float32)
        self._check(self.lib.rawdtw_score_batch(self._ctx, _ptr(jobs), len(jobs), _ptr(ev), len(ev), _ptr(out)))
        return out

    def traceback_batch(self, jobs, events):
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        ev = _f32(events)
        caps = jobs["n"].astype(np.uint64) + jobs["m"].astype(np.uint64) - 1
        off = np.zeros(len(jobs) + 1, np.uint64)
        np.cumsum(caps, out=off[1:])
        total = int(off[-1])
        cost = np.empty(len(jobs), np.float32)
        plen = np.zeros(len(jobs), np.uint32)
        pi = np.zeros(max(total, 1), np.uint32)
        pj = np.zeros(max(total, 1), np.uint32)
        pd = np.zeros(max(total, 1), np.float32)
        self._check(
            self.lib.rawdtw_traceback_batch(
                self._ctx, _ptr(jobs), len(jobs), _ptr(ev), len(ev), _ptr(cost), _ptr(off), _ptr(plen),
                _ptr(pi), _ptr(pj), _ptr(pd),
            )
        )
        res = []
        for k in range(len(jobs)):
            s, e = int(off[k]), int(off[k]) + int(plen[k])
            res.append(DtwResult(cost[k], pi[s:e].copy(), pj[s:e].copy(), pd[s:e].copy()))
        return res

    # -- the reference's own function names (src/dtw.hpp:21,25,28) -----------------
    def DTW_global(self, a_values, b_values, exclude_last_element=False) -> np.float32:
        a, b = _f32(a_values), _f32(b_values)
        c = C.c_float()
        self._check(self.lib.rawdtw_dtw_global(self._ctx, _ptr(a), len(a), _ptr(b), len(b),
                                               int(exclude_last_element), C.byref(c)))
        return np.float32(c.value)

    def DTW_global_slantedbanded_antidiagonalwise(self, a_values, b_values, band_radius,
                                                  exclude_last_element=False) -> np.float32:
        a, b = _f32(a_values), _f32(b_values)
        c = C.c_float()
        self._check(self.lib.rawdtw_dtw_global_slantedbanded_antidiagonalwise(
            self._ctx, _ptr(a), len(a), _ptr(b), len(b), int(band_radius), int(exclude_last_element), C.byref(c)))
        return np.float32(c.value)

    def DTW_global_tb(self, a_values, b_values, exclude_last_element=False) -> DtwResult:
        a, b = _f32(a_values), _f32(b_values)
        cap = len(a) + len(b) - 1
        pi = np.zeros(max(cap, 1), np.uint32)
        pj = np.zeros(max(cap, 1), np.uint32)
        pd = np.zeros(max(cap, 1), np.float32)
        c = C.c_float()
        ln = C.c_uint32()
        self._check(self.lib.rawdtw_dtw_global_tb(self._ctx, _ptr(a), len(a), _ptr(b), len(b),
                                                  int(exclude_last_element), C.byref(c), C.byref(ln),
                                                  _ptr(pi), _ptr(pj), _ptr(pd)))
        k = ln.value
        return DtwResult(np.float32(c.value), pi[:k].copy(), pj[:k].copy(), pd[:k].copy())
