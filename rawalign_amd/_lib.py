"""ctypes binding to librawdtw.so (the C ABI declared in include/rawdtw.h)."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LibraryMissing(RuntimeError):
    pass


class RawDTWError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"rawdtw status {status}: {msg}")
        self.status = status


def library_path() -> str:
    return os.environ.get("RAWDTW_LIBRARY", os.path.join(HERE, "librawdtw.so"))


class AlignOpt(C.Structure):
    _fields_ = [
        ("border_constraint", C.c_int),
        ("fill_method", C.c_int),
        ("band_radius_frac", C.c_float),
        ("match_bonus", C.c_float),
        ("min_score", C.c_float),
        ("fused_score", C.c_int),
    ]


class ChainRec(C.Structure):
    """rawdtw_chain_t"""
    _fields_ = [
        ("chaining_score", C.c_float),
        ("alignment_score", C.c_float),
        ("reference_sequence_index", C.c_uint32),
        ("start_position", C.c_uint32),
        ("end_position", C.c_uint32),
        ("n_anchors", C.c_uint32),
        ("strand", C.c_int32),
        ("mapq", C.c_uint32),
        ("tag", C.c_uint32),
    ]


class SelectOpt(C.Structure):
    """rawdtw_select_opt_t"""
    _fields_ = [
        ("evaluate_chains", C.c_int),
        ("min_bestmap_ratio", C.c_float),
        ("min_meanmap_ratio", C.c_float),
        ("min_chain_anchor", C.c_uint32),
    ]


class ChainOpt(C.Structure):
    """rawdtw_chain_opt_t, defaults of src/roptions.c:13-19"""
    _fields_ = [
        ("max_gap_length", C.c_int),
        ("max_target_gap_length", C.c_int),
        ("chaining_band_length", C.c_int),
        ("max_num_skips", C.c_int),
        ("min_num_anchors", C.c_int),
        ("num_best_chains", C.c_int),
        ("min_chaining_score", C.c_float),
        ("e", C.c_int),
        ("disable_score_filtering", C.c_int),
    ]


class ChainOut(C.Structure):
    _fields_ = [("chaining_score", C.c_float), ("start_position", C.c_uint32), ("end_position", C.c_uint32),
                ("n_anchors", C.c_uint32)]


class MapperOpt(C.Structure):
    """rawdtw_mapper_opt_t"""
    _fields_ = [
        ("flag", C.c_int), ("align", AlignOpt), ("chain", ChainOpt),
        ("min_bestmap_ratio", C.c_float), ("min_meanmap_ratio", C.c_float), ("min_chain_anchor", C.c_uint32),
        ("bp_per_sec", C.c_uint32), ("sample_rate", C.c_uint32), ("chunk_size", C.c_uint32), ("max_num_chunk", C.c_uint32),
        ("slot_events", C.c_uint32), ("max_reads", C.c_uint32), ("carry", C.c_int), ("min_events", C.c_uint32),
        ("threads", C.c_int), ("groups", C.c_int), ("device_chain", C.c_int),
    ]


class SeedHit(C.Structure):
    """rawdtw_seed_hit_t"""
    _fields_ = [("ref_seq", C.c_uint32), ("strand", C.c_int32), ("target_position", C.c_uint32), ("query_position", C.c_uint32)]


# rawdtw_scorer_fn (rawdtw_mapper_set_scorer)
SCORER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                        C.c_void_p, C.c_void_p)


class PlanInfo(C.Structure):
    _fields_ = [
        ("n_jobs", C.c_uint64),
        ("cells", C.c_uint64),
        ("algorithmic_bytes", C.c_uint64),
        ("n_lane_jobs", C.c_uint64),
        ("n_wave_band_jobs", C.c_uint64),
        ("n_full_jobs", C.c_uint64),
        ("workspace_bytes", C.c_uint64),
        ("n_launches", C.c_uint32),
    ]


# every symbol include/rawdtw.h declares: name -> (restype, argtypes)
VP, U64, U32, I32, F32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float
SYMBOLS = {
    "rawdtw_abi_version": (I32, []),
    "rawdtw_device_count": (I32, [C.POINTER(I32)]),
    "rawdtw_create": (I32, [I32, C.POINTER(VP)]),
    "rawdtw_destroy": (I32, [VP]),
    "rawdtw_last_error": (C.c_char_p, [VP]),
    "rawdtw_status_string": (C.c_char_p, [I32]),
    "rawdtw_sync": (I32, [VP]),
    "rawdtw_set_option": (I32, [VP, C.c_char_p, C.c_int64]),
    "rawdtw_stream": (I32, [VP, C.POINTER(VP)]),
    "rawdtw_upload_reference": (I32, [VP, U32, VP, VP, VP]),
    "rawdtw_reference_offset": (I32, [VP, U32, I32, C.POINTER(U64)]),
    "rawdtw_set_reference_device": (I32, [VP, VP, U64]),
    "rawdtw_share_reference": (I32, [VP, VP]),
    "rawdtw_index_open": (I32, [C.c_char_p, C.POINTER(VP)]),
    "rawdtw_index_info": (I32, [VP, C.POINTER(U32), VP]),
    "rawdtw_index_seq": (I32, [VP, U32, C.POINTER(C.c_char_p), C.POINTER(U32)]),
    "rawdtw_index_upload": (I32, [VP, VP]),
    "rawdtw_index_read_signal": (I32, [VP, U32, I32, VP]),
    "rawdtw_index_close": (I32, [VP]),
    "rawdtw_upload_events": (I32, [VP, VP, U64]),
    "rawdtw_set_events_device": (I32, [VP, VP, U64]),
    "rawdtw_events_reserve": (I32, [VP, U64]),
    "rawdtw_events_append": (I32, [VP, VP, U64, U32, VP, VP]),
    "rawdtw_host_alloc": (I32, [U64, C.POINTER(VP)]),
    "rawdtw_host_free": (I32, [VP]),
    "rawdtw_host_is_page_locked": (I32, [VP]),
    "rawdtw_score_batch": (I32, [VP, VP, U64, VP, U64, VP]),
    "rawdtw_plan_create": (I32, [VP, VP, U64, C.POINTER(VP)]),
    "rawdtw_plan_info": (I32, [VP, C.POINTER(PlanInfo)]),
    "rawdtw_plan_dry_run": (I32, [U64, U64, VP, U64, C.c_int, VP, VP, U32, C.POINTER(PlanInfo), C.POINTER(U64), VP, U32]),
    "rawdtw_plan_run": (I32, [VP, VP]),
    "rawdtw_plan_fetch": (I32, [VP, VP, VP]),
    "rawdtw_plan_device_costs": (I32, [VP, C.POINTER(VP), C.POINTER(VP)]),
    "rawdtw_plan_run_timed": (I32, [VP, VP, VP, VP, U32]),
    "rawdtw_plan_destroy": (I32, [VP]),
    "rawdtw_traceback_batch": (I32, [VP, VP, U64, VP, U64, VP, VP, VP, VP, VP, VP]),
    "rawdtw_traceback_batch_steps": (I32, [VP, VP, U64, VP, U64, VP, VP, VP, VP, VP]),
    "rawdtw_dtw_global": (I32, [VP, VP, U32, VP, U32, I32, C.POINTER(F32)]),
    "rawdtw_dtw_global_slantedbanded_antidiagonalwise": (I32, [VP, VP, U32, VP, U32, I32, I32, C.POINTER(F32)]),
    "rawdtw_dtw_global_tb": (I32, [VP, VP, U32, VP, U32, I32, C.POINTER(F32), C.POINTER(U32), VP, VP, VP]),
    "rawdtw_chain_job_count": (U32, [C.POINTER(AlignOpt), U32]),
    "rawdtw_chain_build_jobs": (I32, [C.POINTER(AlignOpt), VP, U32, U64, U32, I32, VP]),
    "rawdtw_chain_replay": (F32, [C.POINTER(AlignOpt), VP, U32, VP, F32]),
    "rawdtw_read_replay": (U32, [C.POINTER(AlignOpt), U32, VP, VP, VP, VP, VP, VP]),
    "rawdtw_gen_primary_chains": (U32, [VP, U32, C.POINTER(SelectOpt), VP]),
    "rawdtw_is_mapped_with_high_confidence": (I32, [VP, U32, C.POINTER(SelectOpt)]),
    "rawdtw_find_outlier": (F32, [VP, U32, U32]),
    "rawdtw_find_outlier_contracted": (F32, [VP, U32, U32]),
    "rawdtw_chain_anchors": (I32, [C.POINTER(ChainOpt), VP, U32, C.POINTER(F32), VP, VP, VP, U32, U64]),
    "rawdtw_sort_by_chaining_score": (I32, [VP, U32, VP]),
    "rawdtw_batch_build_jobs": (I32, [C.POINTER(AlignOpt), U64, VP, VP, VP, VP, VP, VP, U64, C.POINTER(U64)]),
    "rawdtw_batch_create": (I32, [VP, C.POINTER(AlignOpt), U64, VP, VP, VP, VP, VP, C.POINTER(VP)]),
    "rawdtw_batch_verify_plan": (I32, [VP, VP, VP, U64, C.POINTER(C.c_int), VP, U32]),
    "rawdtw_batch_info": (I32, [VP, C.POINTER(PlanInfo), C.POINTER(U64)]),
    "rawdtw_batch_run": (I32, [VP, VP]),
    "rawdtw_batch_run_timed": (I32, [VP, VP, VP, VP, U32, C.POINTER(U32)]),
    "rawdtw_batch_run_reps": (I32, [VP, VP, U32, VP, VP, U32, C.POINTER(U32)]),
    "rawdtw_batch_enqueue": (I32, [VP, VP, I32]),
    "rawdtw_batch_collect": (I32, [VP, VP, VP, VP, U32, C.POINTER(U32), C.POINTER(U32)]),
    "rawdtw_batch_launch_stats": (I32, [VP, U32, C.POINTER(U32), C.POINTER(I32), C.POINTER(U64), C.POINTER(U64), C.POINTER(U64)]),
    "rawdtw_batch_fetch": (I32, [VP, VP, VP, VP, VP]),
    "rawdtw_batch_plan_ms": (I32, [VP, VP, C.POINTER(F32)]),
    "rawdtw_batch_wide_ms": (I32, [VP, VP, C.POINTER(F32)]),
    "rawdtw_batch_stream_counters": (I32, [VP, VP, VP, U32, VP]),
    "rawdtw_batch_destroy": (I32, [VP]),
    "rawdtw_batch_submit": (I32, [VP, VP, U64, VP, VP, VP, VP, VP, VP]),
    "rawdtw_batch_submit_device": (I32, [VP, VP, U64, VP, VP, VP, VP, VP, VP]),
    "rawdtw_chain_round": (I32, [VP, C.POINTER(ChainOpt), U64, VP, VP, VP, U32, VP, VP, VP, VP, U64, VP, VP, VP, VP]),
    "rawdtw_chain_round_begin": (I32, [VP, C.POINTER(ChainOpt), U64, VP, VP, VP, U32, VP, VP, VP, VP, U64, VP]),
    "rawdtw_chain_round_end": (I32, [VP, VP, VP, VP]),
    "rawdtw_batch_fetch_destroy": (I32, [VP, VP, VP, VP]),
    "rawdtw_batch_submit_carry": (I32, [VP, VP, U64, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "rawdtw_batch_can_carry": (I32, [VP, VP, VP]),
    "rawdtw_batch_round_stats": (I32, [VP, VP, VP, VP]),
    "rawdtw_round_match_chains": (I32, [U64, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "rawdtw_batch_stream_counter_index": (I32, [C.c_char_p]),
    "rawdtw_mapper_create": (I32, [VP, VP, U32, VP, VP, VP]),
    "rawdtw_mapper_add_read": (I32, [VP, C.c_char_p, U32, U32, VP]),
    "rawdtw_mapper_round": (I32, [VP, U32, VP, VP, VP, VP, VP]),
    "rawdtw_mapper_read_state": (I32, [VP, U32, VP, VP]),
    "rawdtw_mapper_finish": (I32, [VP]),
    "rawdtw_mapper_paf": (I32, [VP, U32, VP, U32, VP]),
    "rawdtw_mapper_log": (I32, [VP, VP]),
    "rawdtw_mapper_stats": (I32, [VP, VP, VP, VP]),
    "rawdtw_mapper_last_error": (C.c_char_p, [VP]),
    "rawdtw_mapper_destroy": (I32, [VP]),
    "rawdtw_mapper_release_read": (I32, [VP, U32]),
    "rawdtw_mapper_timing": (I32, [VP, VP]),
    "rawdtw_mapper_set_scorer": (I32, [VP, VP, VP]),
    "rawdtw_context_device": (I32, [VP, C.POINTER(I32)]),
    "rawdtw_anchors_pack": (I32, [U64, VP, VP, VP, VP, VP, VP, U64, VP]),
    "rawdtw_anchors_unpack": (I32, [U64, VP, VP, VP, VP, VP, U64, VP]),
    "rawdtw_batch_submit_compact": (I32, [VP, VP, U64, VP, VP, VP, VP, VP, VP, U64, VP, VP, VP]),
    "rawdtw_traceback_timing": (I32, [VP, VP, VP, VP, VP]),
    "rawdtw_batch_replay": (I32, [C.POINTER(AlignOpt), U64, VP, VP, VP, VP, VP, VP, VP]),
}


def load_library():
    """Load librawdtw.so; raises LibraryMissing (never falls back to anything else)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise LibraryMissing(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C rawalign_amd/csrc` (hipcc, --offload-arch=gfx950)"
        )
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
