"""ctypes binding of the C ABI in include/abyss_amd.h (libabyss_amd.so)."""
from __future__ import annotations

import ctypes as C
import os

from . import build

ABG_OK = 0
ABG_EINVAL, ABG_ENODEV, ABG_ENOMEM, ABG_EINTERNAL, ABG_EAGAIN = -1, -2, -3, -4, -5


class Params(C.Structure):
    _fields_ = [
        ("k", C.c_uint32), ("num_hashes", C.c_uint32), ("min_cov", C.c_uint32), ("trim", C.c_uint32),
        ("bloom_bytes", C.c_uint64), ("counters", C.c_uint64), ("spaced_seed", C.c_char_p),
        ("device", C.c_int32), ("verbose", C.c_int32), ("insert_batch_kmers", C.c_uint64),
        ("claim_log2", C.c_uint32), ("walk_slots", C.c_uint32), ("wtab_log2", C.c_uint32),
        ("cascade_levels", C.c_uint32), ("slice_filter", C.c_uint32), ("reserved_", C.c_uint32 * 5),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("solid_reads", "visited_reads", "reads_processed", "bases_assembled", "next_contig_id")]


class Contig(C.Structure):
    _fields_ = [
        ("contig_id", C.c_uint64), ("read_index", C.c_uint64), ("seq", C.c_char_p),
        ("length", C.c_uint32), ("coverage", C.c_uint32), ("redundant", C.c_int32),
        ("left_ext", C.c_uint32), ("right_ext", C.c_uint32), ("left_code", C.c_int32),
        ("right_code", C.c_int32), ("seed_pos", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("insert_rounds", "walk_rounds", "candidates", "walked", "rewalked", "commit_breaks",
                 "commit_rounds", "generated", "bulk_calls", "bulk_steps", "lin_steps", "guide_slots", "chain_steps", "batch_cuts", "overflows", "memo_hits", "memo_adds", "tiled_ops", "tiled_pending", "tile_overflows", "cls_covered_reads", "archive_bases", "cls_decided_reads", "counter_bytes_held")]


CONTIG_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Contig))
TEXT_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_char), C.c_uint64)

_lib = None


def symbols():
    """Every entry point include/abyss_amd.h declares."""
    return [
        "abg_params_init", "abg_create", "abg_destroy", "abg_last_error", "abg_reset", "abg_filter_size",
        "abg_load_seqs", "abg_load_seqs_v", "abg_keep_reads", "abg_assemble_kept", "abg_load_packed", "abg_counting_stats", "abg_counters_export",
        "abg_counters_import", "abg_visited_export", "abg_visited_import", "abg_assemble_seqs", "abg_assemble_seqs_v",
        "abg_assemble_packed", "abg_cascade_export", "abg_get_counters", "abg_set_counters", "abg_hash_seq",
        "abg_contains_seq",
        "abg_attach_comm", "abg_share_reads", "abg_rccl_unique_id", "abg_rccl_comm_create", "abg_rccl_comm_destroy",
        "abg_dev_copy", "abg_dev_alloc", "abg_dev_free", "abg_output_graph_seqs",
        "abg_profile_enable", "abg_profile_reset", "abg_profile_get", "abg_get_stats",
        "abg_overlap_create", "abg_overlap_destroy", "abg_overlap_last_error", "abg_overlap_join", "abg_overlap_edges",
        "abg_overlap_profile", "abg_overlap_profile_get",
        "abg_rr_create", "abg_rr_destroy", "abg_rr_last_error", "abg_rr_bytes", "abg_rr_clear", "abg_rr_insert_seqs",
        "abg_rr_contains_seqs", "abg_rr_popcount", "abg_rr_export", "abg_rr_sync", "abg_rr_profile", "abg_rr_profile_get",
    ]


def load(path: str | None = None):
    """Load libabyss_amd.so (building it in-tree first if it is missing or stale)."""
    global _lib
    if _lib is not None:
        return _lib
    if path is None:
        path = os.environ.get("ABG_LIB") or build.LIB  # ABG_LIB: a differently tuned build (experiments)
        if not os.path.exists(path):
            build.build_lib()
    lib = C.CDLL(path)
    vp, u64p, u8p = C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p
    lib.abg_params_init.argtypes = [C.POINTER(Params)]
    lib.abg_params_init.restype = None
    lib.abg_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    lib.abg_destroy.argtypes = [vp]
    lib.abg_destroy.restype = None
    lib.abg_last_error.argtypes = [vp]
    lib.abg_last_error.restype = C.c_char_p
    lib.abg_reset.argtypes = [vp]
    lib.abg_contains_seq.argtypes = [vp, C.c_char_p, C.c_uint64, vp, vp, C.c_uint64, u64p]
    lib.abg_filter_size.argtypes = [vp, u64p]
    lib.abg_load_seqs.argtypes = [vp, C.c_char_p, vp, C.c_uint64]
    lib.abg_load_packed.argtypes = [vp, vp, vp, vp, C.c_uint64]
    lib.abg_load_seqs_v.argtypes = [vp, C.c_uint32, vp, vp, vp]
    lib.abg_keep_reads.argtypes = [vp, C.c_int, C.c_uint64]
    lib.abg_assemble_kept.argtypes = [vp, vp, CONTIG_CB, vp]
    lib.abg_counting_stats.argtypes = [vp, u64p, u64p]
    lib.abg_counters_export.argtypes = [vp, u8p]
    lib.abg_counters_import.argtypes = [vp, u8p]
    lib.abg_visited_export.argtypes = [vp, u8p]
    lib.abg_visited_import.argtypes = [vp, u8p]
    lib.abg_assemble_seqs.argtypes = [vp, C.c_char_p, vp, C.c_uint64, vp, CONTIG_CB, vp]
    lib.abg_assemble_seqs_v.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, CONTIG_CB, vp]
    lib.abg_assemble_packed.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, CONTIG_CB, vp]
    lib.abg_cascade_export.argtypes = [vp, C.c_uint32, u8p]
    lib.abg_get_counters.argtypes = [vp, C.POINTER(Counters)]
    lib.abg_set_counters.argtypes = [vp, C.POINTER(Counters)]
    lib.abg_hash_seq.argtypes = [vp, C.c_char_p, C.c_uint64, vp, vp, C.c_uint64, u64p]
    lib.abg_profile_enable.argtypes = [vp, C.c_int]
    lib.abg_profile_reset.argtypes = [vp]
    lib.abg_profile_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), u64p]
    lib.abg_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.abg_attach_comm.argtypes = [vp, vp]
    lib.abg_share_reads.argtypes = [vp, vp, vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), u64p]
    lib.abg_rccl_unique_id.argtypes = [vp]
    lib.abg_rccl_comm_create.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.abg_rccl_comm_destroy.argtypes = [vp]
    lib.abg_dev_copy.argtypes = [vp, vp, vp, C.c_uint64, C.c_int32]
    lib.abg_output_graph_seqs.argtypes = [vp, C.c_char_p, vp, C.c_uint64, TEXT_CB, vp, u64p, u64p]
    lib.abg_dev_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    lib.abg_dev_free.argtypes = [vp, vp]
    lib.abg_overlap_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.abg_overlap_destroy.argtypes = [vp]
    lib.abg_overlap_destroy.restype = None
    lib.abg_overlap_last_error.argtypes = [vp]
    lib.abg_overlap_last_error.restype = C.c_char_p
    lib.abg_overlap_join.argtypes = [vp, C.c_uint32, C.c_uint64, vp, vp, C.c_int, u64p]
    lib.abg_overlap_edges.argtypes = [vp, vp, vp]
    lib.abg_overlap_profile.argtypes = [vp, C.c_int]
    lib.abg_overlap_profile_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), u64p]
    lib.abg_rr_create.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    lib.abg_rr_destroy.argtypes = [vp]
    lib.abg_rr_destroy.restype = None
    lib.abg_rr_last_error.argtypes = [vp]
    lib.abg_rr_last_error.restype = C.c_char_p
    lib.abg_rr_bytes.argtypes = [vp, u64p]
    lib.abg_rr_clear.argtypes = [vp]
    lib.abg_rr_insert_seqs.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, vp, C.c_uint32, u64p]
    lib.abg_rr_contains_seqs.argtypes = [vp, vp, vp, C.c_uint64, vp]
    lib.abg_rr_popcount.argtypes = [vp, u64p]
    lib.abg_rr_export.argtypes = [vp, vp]
    lib.abg_rr_sync.argtypes = [vp]
    lib.abg_rr_profile.argtypes = [vp, C.c_int]
    lib.abg_rr_profile_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), u64p]
    _lib = lib
    return lib
