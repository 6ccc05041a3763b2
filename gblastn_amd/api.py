"""Host-side mirror of the reference's preliminary-search interface, over the C ABI.

Names follow the reference's operator surface for this path:
  BlastSeqSrc       ~ the BlastSeqSrc / CSearchDatabase handle the engine iterates
                      (API/seqsrc_seqdb.cpp): here one HBM-resident shard
  BlastPrelimSearch ~ CBlastPrelimSearch(query_factory, options, dbinfo).Run()
                      (API/prelim_stage.cpp:192-308), which calls
                      Blast_gpu_RunPreliminarySearchWithInterrupt
                      (GB/gpu_blastn_pre_search_engine.cpp:1466-1563)
The HIP extension is mandatory: import fails loudly if libgblastn_amd.so is absent.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("GBN_AMD_LIB") or os.path.join(_HERE, "libgblastn_amd.so")   # GBN_AMD_LIB: A/B builds (tools/build_variant.sh)


class GbnOptions(C.Structure):
    _fields_ = [("word_size", C.c_int32), ("reward", C.c_int32), ("penalty", C.c_int32),
                ("gap_open", C.c_int32), ("gap_extend", C.c_int32), ("greedy", C.c_int32),
                ("xdrop_ungap_bits", C.c_double), ("gap_trigger_bits", C.c_double),
                ("xdrop_gap_bits", C.c_double), ("xdrop_gap_final_bits", C.c_double),
                ("evalue", C.c_double), ("min_diag_separation", C.c_int32),
                ("hitlist_size", C.c_int32), ("cutoff_score", C.c_int32),
                ("lut11_gblastn_rule", C.c_int32), ("db_length", C.c_int64),
                ("db_num_seqs", C.c_int32)]


class GbnContext(C.Structure):
    _fields_ = [("query_offset", C.c_int32), ("query_length", C.c_int32), ("frame", C.c_int32),
                ("query_index", C.c_int32), ("is_valid", C.c_int32),
                ("length_adjustment", C.c_int32), ("eff_searchsp", C.c_int64),
                ("lambda_u", C.c_double), ("K_u", C.c_double), ("logK_u", C.c_double),
                ("H_u", C.c_double), ("x_dropoff", C.c_int32), ("cutoff_score", C.c_int32),
                ("reduced_cutoff", C.c_int32), ("gap_cutoff_score", C.c_int32),
                ("gap_cutoff_score_max", C.c_int32)]


class GbnDiagnostics(C.Structure):
    _fields_ = [("lookup_hits", C.c_int64), ("init_extends", C.c_int64),
                ("good_init_extends", C.c_int64), ("gapped_extensions", C.c_int64),
                ("good_extensions", C.c_int64), ("seqs_passed", C.c_int64), ("seeds", C.c_int64),
                ("scan_kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("scan_launches", C.c_int64), ("subject_bases_scanned", C.c_int64),
                ("bin_kernel_ms", C.c_double), ("probe_kernel_ms", C.c_double),
                ("rare_kernel_ms", C.c_double), ("scan_stage_ms", C.c_double),
                ("seed_stage_ms", C.c_double), ("gapped_stage_ms", C.c_double),
                ("host_stage_ms", C.c_double), ("kernel_ms", C.c_double * 8),
                ("ranges", C.c_int64), ("scan_rescans", C.c_int64), ("direct_ranges", C.c_int64), ("library_sorts", C.c_int64)]
    KERNEL_CLASSES = ["seed keys", "seed order (radix sort / seed_order kernels)", "seed_ext_ck_kernel + seed_exact_kernel", "diag_replay_kernel", "diag_ungapped_kernel",
                      "dynprog_lane_kernel", "dynprog_wave_kernel / greedy_wave_kernel", "dynprog_kernel / greedy_kernel"]


HSP_DT = np.dtype([("oid", "<i4"), ("context", "<i4"), ("q_offset", "<i4"), ("q_end", "<i4"),
                   ("q_gapped_start", "<i4"), ("s_offset", "<i4"), ("s_end", "<i4"),
                   ("s_gapped_start", "<i4"), ("score", "<i4"), ("pad_", "<i4"),
                   ("evalue", "<f8")])
TB_DT = np.dtype([("oid", "<i4"), ("context", "<i4"), ("q_offset", "<i4"), ("q_end", "<i4"),
                  ("q_gapped_start", "<i4"), ("s_offset", "<i4"), ("s_end", "<i4"),
                  ("s_gapped_start", "<i4"), ("score", "<i4"), ("pad_", "<i4"), ("evalue", "<f8"),
                  ("num_ident", "<i4"), ("align_length", "<i4"), ("gaps", "<i4"), ("gap_opens", "<i4"),
                  ("ops_first", "<i8"), ("ops_count", "<i4"), ("pad2_", "<i4"), ("bit_score", "<f8")])
SEED_DT = np.dtype([("oid", "<i4"), ("s_off", "<i4"), ("q_off", "<i4"), ("pad_", "<i4")])
IHIT_DT = np.dtype([("oid", "<i4"), ("q_off", "<i4"), ("s_off", "<i4"), ("q_start", "<i4"),
                    ("s_start", "<i4"), ("length", "<i4"), ("score", "<i4"), ("pad_", "<i4")])

EXPORTS = ["gbn_init", "gbn_release", "gbn_release_db_memory", "gbn_debug_check_guards", "gbn_device_count", "gbn_use_device", "gbn_current_device", "gbn_db_device", "gbn_shard_builder_add_oid", "gbn_default_options",
           "gbn_db_new", "gbn_db_new_streamed", "gbn_db_free", "gbn_db_total_bases", "gbn_db_num_seqs", "gbn_synth_fill", "gbn_synth_skew",
           "gbn_batch_new", "gbn_batch_new_ex", "gbn_batch_new_masked", "gbn_dust_mask", "gbn_batch_free", "gbn_batch_num_contexts", "gbn_batch_contexts",
           "gbn_batch_lut_type", "gbn_batch_lut_width", "gbn_batch_scan_step", "gbn_batch_scan_path",
           "gbn_batch_diag_container", "gbn_batch_gap_x_dropoff", "gbn_results_new",
           "gbn_results_free", "gbn_results_clear", "gbn_results_num_hsps", "gbn_results_hsps",
           "gbn_results_num_seeds", "gbn_results_seeds", "gbn_results_num_init_hits",
           "gbn_results_init_hits", "gbn_prelim_search", "gbn_scan_only", "gbn_last_error",
           "gbn_launch_scan_seed", "gbn_launch_ungapped", "gbn_launch_gapped", "gbn_batch_karlin_gapped",
           "gbn_prelim_search_begin", "gbn_prelim_search_end", "gbn_prelim_hitlist_size", "gbn_collector_new", "gbn_collector_free", "gbn_collector_write",
           "gbn_collector_close", "gbn_collector_num_lists", "gbn_collector_list_starts",
           "gbn_collector_list_queries", "gbn_collector_num_hsps", "gbn_collector_hsps",
           "gbn_blastdb_open", "gbn_blastdb_close", "gbn_blastdb_num_volumes", "gbn_blastdb_num_seqs",
           "gbn_blastdb_total_length", "gbn_blastdb_max_length", "gbn_blastdb_stat_num_seqs",
           "gbn_blastdb_stat_length", "gbn_blastdb_title", "gbn_blastdb_volume_range", "gbn_blastdb_seq_length",
           "gbn_blastdb_get_ncbi2na", "gbn_blastdb_num_ambiguities", "gbn_blastdb_get_ambiguities",
           "gbn_blastdb_get_blastna", "gbn_blastdb_load_shard",
           "gbn_traceback_new", "gbn_traceback_free", "gbn_traceback_run", "gbn_traceback_num_hsps", "gbn_traceback_hsps",
           "gbn_traceback_ops", "gbn_traceback_op_lengths", "gbn_traceback_query_starts",
           "gbn_pipeline_new", "gbn_pipeline_free", "gbn_pipeline_submit", "gbn_pipeline_finish", "gbn_pipeline_next",
           "gbn_pipeline_diagnostics",
           "gbn_batch_scan_params", "gbn_batch_ext_params", "gbn_batch_gap_params", "gbn_batch_diag_layout",
           "gbn_prelim_search_lists", "gbn_db_cache_find", "gbn_db_cache_insert", "gbn_block_cache_find", "gbn_block_cache_insert",
           "gbn_debug_db_bytes_uploaded", "gbn_debug_seed_order", "gbn_debug_bin_ahead_hits", "gbn_debug_bin_ahead_misses",
           "gbn_record_cache_set_limit", "gbn_record_cache_stats", "gbn_record_cache_invalidate", "gbn_db_prepare_records", "gbn_block_view", "gbn_results_emit_lists", "gbn_debug_counting_sink",
           "gbn_set_max_dbseq_len", "gbn_db_set_ambiguities", "gbn_traceback_merge", "gbn_shard_builder_new", "gbn_shard_builder_add", "gbn_shard_builder_finish", "gbn_shard_builder_free"]

# ---- include/gblastn_amd_kernels.h: parameter blocks of the gbn_launch_* entry points (device pointers as integers)
_P, _I, _L, _U, _UL = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
TILE_DT = np.dtype([("subj", "<i4"), ("first_pos", "<i4"), ("npos", "<i4"), ("off16", "<i4")])
DEV_SEED_DT = np.dtype([("subj", "<i4"), ("s_scan", "<i4"), ("q_pos", "<i4"), ("ext_left", "<i4")])
DEV_IHIT_DT = np.dtype([("subj", "<i4"), ("q_off", "<i4"), ("s_off", "<i4"), ("q_start", "<i4"), ("s_start", "<i4"),
                        ("length", "<i4"), ("score", "<i4"), ("seq", "<u4")])
DEV_GAPPED_DT = np.dtype([("q_start", "<i4"), ("q_stop", "<i4"), ("s_start", "<i4"), ("s_stop", "<i4"), ("score", "<i4"),
                          ("seed_q", "<i4"), ("seed_s", "<i4"), ("context", "<i4")])
GBN_TILE_POS = 2048


class GbnScanParams(C.Structure):
    _fields_ = [("db", _P), ("byte_off", _P), ("len", _P), ("tiles", _P), ("ntiles", _L),
                ("pv", _P), ("cellw", _P), ("cell_start", _P), ("ent", _P), ("ncells", _L),
                ("lut", C.c_int), ("word", C.c_int), ("step", C.c_int), ("mode", C.c_int), ("fl", C.c_int), ("fr", C.c_int),
                ("q8", _P), ("qlen", _I), ("ctx_off", _P), ("ctx_len", _P), ("nctx", _I),
                ("seeds", _P), ("seed_count", _P), ("seed_cap", _UL), ("raw_hits", _P), ("pvx", _P), ("pstart", _P)]


class GbnExtParams(C.Structure):
    _fields_ = [("db", _P), ("byte_off", _P), ("len", _P), ("seeds", _P), ("idx", _P), ("key_group", _P), ("n", _L),
                ("q8", _P), ("qlen", _I), ("ctx_off", _P), ("ctx_len", _P), ("ctx_xdrop", _P), ("ctx_cutoff", _P),
                ("ctx_reduced", _P), ("nctx", _I), ("matrix", _P), ("score_table", _P), ("word", C.c_int),
                ("container_hash", C.c_int), ("cell_diag", _P), ("cell_level", _P), ("cell_start", _P), ("ent", _P),
                ("cell_mask", _U), ("lut", C.c_int), ("masked", C.c_int), ("q2", _P), ("qinv", _P),
                ("run_heads", _P), ("run_count", _P), ("group_bits", _I),
                ("ihits", _P), ("ihit_count", _P), ("ihit_cap", _UL), ("ctx_hint", _P), ("ctx_hint_shift", _I), ("ext_rec", _P),
                ("ck_shift", _I), ("ck_s_bits", _I), ("ck_qh_bits", _I), ("ck_q_bits", _I), ("ck_q_desc", _I), ("ck_subj_base", _I), ("ck_vbits", _I), ("q4", _P), ("q4_plane", _L), ("q4_origin", _I), ("ctx_blk", _P), ("ctx_pack", _P), ("exact_list", _P), ("exact_count", _P)]


class GbnGapParams(C.Structure):
    _fields_ = [("db", _P), ("byte_off", _P), ("len", _P), ("ihits", _P), ("first", _L), ("n", _L),
                ("q8", _P), ("ctx_off", _P), ("ctx_len", _P), ("nctx", _I), ("q2", _P), ("qinv", _P), ("matrix", _P),
                ("reward", _I), ("penalty", _I), ("gap_open", _I), ("gap_extend", _I), ("xdrop", _I),
                ("scratch", _P), ("scratch_per_thread", _I), ("row_len", _I), ("out", _P),
                ("max_blocks", _I), ("redo_only", _I)]


GbnHspListFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32)

_LIB = None


def lib():
    """Load the HIP extension; there is no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise ImportError("gblastn_amd: %s is missing -- build it with "
                              "`python -c 'import __graft_entry__ as g; g.build()'`" % _SO)
        L = C.CDLL(_SO)
        L.gbn_last_error.restype = C.c_char_p
        L.gbn_default_options.argtypes = [C.POINTER(GbnOptions), C.c_int]
        L.gbn_init.argtypes = [C.c_int, C.c_int]
        L.gbn_use_device.argtypes = [C.c_int]; L.gbn_db_device.argtypes = [C.c_void_p]
        L.gbn_shard_builder_add_oid.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.gbn_db_new.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_int32, C.c_int]
        L.gbn_db_free.argtypes = [C.c_void_p]
        for f, t in (("gbn_batch_scan_params", GbnScanParams), ("gbn_batch_ext_params", GbnExtParams), ("gbn_batch_gap_params", GbnGapParams)):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.POINTER(t)]
        L.gbn_batch_diag_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.gbn_launch_scan_seed.argtypes = [C.POINTER(GbnScanParams), C.c_int, C.c_void_p]
        L.gbn_launch_ungapped.argtypes = [C.POINTER(GbnExtParams), C.c_void_p]
        L.gbn_launch_gapped.argtypes = [C.POINTER(GbnGapParams), C.c_int, C.c_void_p]
        L.gbn_prelim_search_lists.argtypes = [C.c_void_p, C.c_void_p, GbnHspListFn, C.c_void_p, C.POINTER(GbnDiagnostics), C.c_void_p, C.c_void_p]
        L.gbn_db_cache_find.restype = C.c_void_p; L.gbn_db_cache_find.argtypes = [C.c_void_p]
        L.gbn_db_cache_insert.argtypes = [C.c_void_p, C.c_void_p]
        L.gbn_block_cache_find.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        L.gbn_block_cache_insert.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
        L.gbn_debug_db_bytes_uploaded.restype = C.c_longlong; L.gbn_debug_db_bytes_uploaded.argtypes = []
        L.gbn_debug_bin_ahead_hits.restype = C.c_longlong; L.gbn_debug_bin_ahead_hits.argtypes = []
        if hasattr(L, "gbn_record_cache_set_limit"):            # (an older build loaded through GBN_AMD_LIB for an A/B has none of these)
            L.gbn_debug_bin_ahead_misses.restype = C.c_longlong; L.gbn_debug_bin_ahead_misses.argtypes = []
            L.gbn_record_cache_set_limit.argtypes = [C.c_longlong]
            L.gbn_record_cache_stats.argtypes = [C.POINTER(C.c_longlong), C.c_int]
            L.gbn_block_view.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]
            L.gbn_db_prepare_records.argtypes = [C.c_void_p, C.POINTER(GbnOptions), C.c_int32, C.POINTER(C.c_int32)]
            L.gbn_results_emit_lists.argtypes = [C.c_void_p, GbnHspListFn, C.c_void_p]
        L.gbn_debug_seed_order.restype = C.c_int
        L.gbn_debug_seed_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_int32,
                                           C.c_int, C.c_void_p, C.POINTER(C.c_int64)]
        L.gbn_shard_builder_new.argtypes = [C.POINTER(C.c_void_p), C.c_int32]
        L.gbn_shard_builder_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.gbn_shard_builder_finish.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.gbn_shard_builder_free.argtypes = [C.c_void_p]
        L.gbn_db_total_bases.restype = C.c_int64; L.gbn_db_total_bases.argtypes = [C.c_void_p]
        L.gbn_db_num_seqs.restype = C.c_int32; L.gbn_db_num_seqs.argtypes = [C.c_void_p]
        L.gbn_synth_fill.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p]
        L.gbn_synth_skew.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_uint64, C.c_void_p]
        L.gbn_batch_new.argtypes = [C.POINTER(C.c_void_p), C.POINTER(GbnOptions), C.c_int32,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
        L.gbn_batch_new_ex.argtypes = [C.POINTER(C.c_void_p), C.POINTER(GbnOptions), C.c_int32,
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int]
        L.gbn_batch_new_masked.argtypes = [C.POINTER(C.c_void_p), C.POINTER(GbnOptions), C.c_int32,
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32,
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]
        L.gbn_dust_mask.restype = C.c_int32
        L.gbn_dust_mask.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        L.gbn_batch_free.argtypes = [C.c_void_p]
        for nm in ["gbn_batch_num_contexts", "gbn_batch_lut_type", "gbn_batch_lut_width",
                   "gbn_batch_scan_step", "gbn_batch_scan_path", "gbn_batch_diag_container", "gbn_batch_gap_x_dropoff"]:
            if nm == "gbn_batch_scan_path" and not hasattr(L, nm):
                continue                                    # (an older build loaded through GBN_AMD_LIB for an A/B)
            getattr(L, nm).restype = C.c_int32; getattr(L, nm).argtypes = [C.c_void_p]
        L.gbn_batch_contexts.restype = C.POINTER(GbnContext); L.gbn_batch_contexts.argtypes = [C.c_void_p]
        L.gbn_results_new.argtypes = [C.POINTER(C.c_void_p)]
        L.gbn_results_free.argtypes = [C.c_void_p]
        L.gbn_results_clear.argtypes = [C.c_void_p]
        for nm in ["gbn_results_num_hsps", "gbn_results_num_seeds", "gbn_results_num_init_hits"]:
            getattr(L, nm).restype = C.c_int64; getattr(L, nm).argtypes = [C.c_void_p]
        for nm in ["gbn_results_hsps", "gbn_results_seeds", "gbn_results_init_hits"]:
            getattr(L, nm).restype = C.c_void_p; getattr(L, nm).argtypes = [C.c_void_p]
        L.gbn_prelim_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(GbnDiagnostics), C.c_int, C.c_void_p, C.c_void_p]
        L.gbn_scan_only.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(GbnDiagnostics)]
        L.gbn_prelim_search_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(GbnDiagnostics),
                                              C.c_void_p, C.c_void_p]
        L.gbn_prelim_search_end.argtypes = [C.c_void_p]
        L.gbn_prelim_hitlist_size.restype = C.c_int32; L.gbn_prelim_hitlist_size.argtypes = [C.c_int32]
        L.gbn_collector_new.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32]
        L.gbn_collector_free.argtypes = [C.c_void_p]
        L.gbn_collector_write.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.gbn_collector_close.argtypes = [C.c_void_p]
        for nm in ["gbn_collector_num_lists", "gbn_collector_num_hsps"]:
            getattr(L, nm).restype = C.c_int64; getattr(L, nm).argtypes = [C.c_void_p]
        for nm in ["gbn_collector_list_starts", "gbn_collector_list_queries", "gbn_collector_hsps"]:
            getattr(L, nm).restype = C.c_void_p; getattr(L, nm).argtypes = [C.c_void_p]
        L.gbn_traceback_new.argtypes = [C.POINTER(C.c_void_p)]
        L.gbn_traceback_free.argtypes = [C.c_void_p]
        L.gbn_traceback_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        L.gbn_traceback_num_hsps.restype = C.c_int64; L.gbn_traceback_num_hsps.argtypes = [C.c_void_p]
        for nm in ["gbn_traceback_hsps", "gbn_traceback_ops", "gbn_traceback_op_lengths", "gbn_traceback_query_starts"]:
            getattr(L, nm).restype = C.c_void_p; getattr(L, nm).argtypes = [C.c_void_p]
        L.gbn_pipeline_new.argtypes = [C.POINTER(C.c_void_p), C.POINTER(GbnOptions), C.c_void_p, C.c_int32, C.c_int, C.c_int]
        L.gbn_pipeline_free.argtypes = [C.c_void_p]
        L.gbn_pipeline_submit.restype = C.c_int64
        L.gbn_pipeline_submit.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.gbn_pipeline_finish.argtypes = [C.c_void_p]
        L.gbn_pipeline_next.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.gbn_pipeline_diagnostics.argtypes = [C.c_void_p, C.POINTER(GbnDiagnostics)]
        L.gbn_blastdb_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
        L.gbn_blastdb_close.argtypes = [C.c_void_p]
        for nm in ["gbn_blastdb_num_volumes", "gbn_blastdb_num_seqs", "gbn_blastdb_max_length", "gbn_blastdb_stat_num_seqs"]:
            getattr(L, nm).restype = C.c_int32; getattr(L, nm).argtypes = [C.c_void_p]
        for nm in ["gbn_blastdb_total_length", "gbn_blastdb_stat_length"]:
            getattr(L, nm).restype = C.c_int64; getattr(L, nm).argtypes = [C.c_void_p]
        L.gbn_blastdb_title.restype = C.c_char_p; L.gbn_blastdb_title.argtypes = [C.c_void_p]
        L.gbn_blastdb_volume_range.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.gbn_blastdb_seq_length.restype = C.c_int32; L.gbn_blastdb_seq_length.argtypes = [C.c_void_p, C.c_int32]
        L.gbn_blastdb_get_ncbi2na.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.gbn_blastdb_num_ambiguities.restype = C.c_int32; L.gbn_blastdb_num_ambiguities.argtypes = [C.c_void_p, C.c_int32]
        L.gbn_blastdb_get_ambiguities.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.gbn_blastdb_get_blastna.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int]
        L.gbn_blastdb_load_shard.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        _LIB = L
    return _LIB


class BlastError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise BlastError("gblastn_amd status %d: %s" % (rc, lib().gbn_last_error().decode()))


def default_options(task="megablast", db_length=0, db_num_seqs=0, **kw):
    o = GbnOptions()
    lib().gbn_default_options(C.byref(o), 1 if task == "megablast" else 0)
    o.db_length = db_length; o.db_num_seqs = db_num_seqs
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def record_cache_set_limit(nbytes=-1):
    """Bytes of scan records the calling thread's device keeps resident ("bin once, probe many"); 0: off, < 0: default."""
    _check(lib().gbn_record_cache_set_limit(int(nbytes)))


def record_cache_invalidate():
    """Every cached record set forgets its records (the buffers stay): the next pass of each key bins again."""
    _check(lib().gbn_record_cache_invalidate())


def granted_cpus():
    """What gbn_host_cpus computes, without the library: hardware threads cut down to the affinity mask and the cgroup's CPU quota.
    For launchers that hand every rank of a node its share (GBN_HOST_CPUS) before the library sizes its pools."""
    import math
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, math.ceil(q / per)))
        except Exception:
            pass
    return n


def share_cpus_among_local_ranks():
    """One process per GPU on a node: every rank sizes its host pools from its share of the CPUs the node (or its container) grants, not
    from all of them.  Call before the first library call of the process (gbn_host_cpus reads GBN_HOST_CPUS once)."""
    lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if lw > 1:
        os.environ.setdefault("GBN_HOST_CPUS", str(max(2, granted_cpus() // lw)))


def host_cpus():
    """CPUs this process may use at a time (hardware threads cut down to the affinity mask and the cgroup's CPU quota):
    gbn_host_cpus; what thread pools on the host are sized from."""
    L = lib()
    L.gbn_host_cpus.restype = C.c_int32; L.gbn_host_cpus.argtypes = []
    return int(L.gbn_host_cpus())


def record_cache_stats():
    v = (C.c_longlong * 14)()
    _check(lib().gbn_record_cache_stats(v, 14))
    return dict(zip(("limit", "bytes", "sets", "hits", "misses", "evictions", "bypassed", "ahead_hits", "prepared",
                     "sorted_sets", "sorted_bytes", "sorts", "sorted_passes", "last_sort_us"), [int(x) for x in v]))


def block_view(blocks):
    """Several resident blocks (BlastSeqSrc) searched as ONE shard; the library caches and owns the view."""
    arr = (C.c_void_p * len(blocks))(*[b._h for b in blocks])
    h = C.c_void_p()
    _check(lib().gbn_block_view(arr, len(blocks), C.byref(h)))
    return BlastSeqSrc(h, keep=list(blocks), owned=False)


def set_max_dbseq_len(n=200000000):
    """MAX_DBSEQ_LEN for shards made from now on (longer sequences are searched in overlapping chunks)"""
    _check(lib().gbn_set_max_dbseq_len(n))


def layout_slab(lengths, front=16, align=16, tail=128):
    """Byte offsets of NCBI2na subjects in one slab (16-byte aligned, padded)."""
    offs = np.zeros(len(lengths), dtype=np.int64)
    pos = front
    for i, n in enumerate(lengths):
        offs[i] = pos
        pos += (int(n) + 3) // 4
        pos = (pos + align - 1) // align * align
    return offs, pos + tail


def dust_masks(queries, level=20, window=64, linker=1):
    """blastn's default query filter: symmetric DUST intervals of every query, as the mask list
    BlastPrelimSearch(masks=...) takes."""
    out = []
    for qi, q in enumerate(queries):
        a = np.ascontiguousarray(q, dtype=np.uint8)
        cap = len(a) // 2 + 4
        f = np.zeros(cap, dtype=np.int32); t = np.zeros(cap, dtype=np.int32)
        n = lib().gbn_dust_mask(a.ctypes.data, len(a), level, window, linker, f.ctypes.data, t.ctypes.data, cap)
        out += [(qi, int(f[i]), int(t[i])) for i in range(n)]
    return out


class BlastSeqSrc:
    """One database shard resident in HBM."""

    def __init__(self, handle, keep=None, owned=True):
        self._h = handle
        self._keep = keep
        self._owned = owned         # False: a view or a block the library's caches own (close() only forgets the handle)

    @classmethod
    def from_packed(cls, subjects, first_oid=0):
        """subjects: list of (packed uint8 array, length in bases)."""
        lens = np.array([n for _, n in subjects], dtype=np.int32)
        offs, total = layout_slab(lens)
        slab = np.zeros(total, dtype=np.uint8)
        for (p, n), o in zip(subjects, offs):
            nb = (n + 3) // 4
            slab[o:o + nb] = np.asarray(p, dtype=np.uint8)[:nb]
        return cls.from_slab(slab, offs, lens, first_oid, is_device=False)

    @classmethod
    def from_slab(cls, slab, byte_off, lens, first_oid=0, is_device=False, keep=None):
        L = lib()
        h = C.c_void_p()
        byte_off = np.ascontiguousarray(byte_off, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        if is_device:
            ptr, nbytes = slab       # (device pointer, size)
        else:
            slab = np.ascontiguousarray(slab, dtype=np.uint8)
            ptr, nbytes = slab.ctypes.data, slab.nbytes
        _check(L.gbn_db_new(C.byref(h), ptr, nbytes, len(lens), byte_off.ctypes.data,
                            lens.ctypes.data, first_oid, 1 if is_device else 0))
        return cls(h, keep)

    @property
    def total_bases(self):
        return lib().gbn_db_total_bases(self._h)

    @property
    def num_seqs(self):
        return lib().gbn_db_num_seqs(self._h)

    def prepare_records(self, options, queries):
        """The scan records a batch of these queries (a QuerySet, or BLASTNA arrays) will want of this shard, binned now,
        asynchronously, underneath the batch's set-up (gbn_db_prepare_records)."""
        qs = queries if isinstance(queries, QuerySet) else QuerySet(queries)
        _check(lib().gbn_db_prepare_records(self._h, C.byref(options), len(qs), qs.lens))

    def close(self):
        if self._h:
            if self._owned:
                lib().gbn_db_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def from_oids(cls, subjects, oids):
        """A block as the shim builds one: subjects appended one by one with their OIDs (gbn_shard_builder_*)."""
        L = lib()
        sb = C.c_void_p()
        _check(L.gbn_shard_builder_new(C.byref(sb), len(subjects)))
        try:
            for (p, n), oid in zip(subjects, oids):
                a = np.ascontiguousarray(p, dtype=np.uint8)
                _check(L.gbn_shard_builder_add_oid(sb, int(oid), a.ctypes.data, int(n)))
            h = C.c_void_p()
            _check(L.gbn_shard_builder_finish(sb, C.byref(h)))
        finally:
            L.gbn_shard_builder_free(sb)
        return cls(h)


class QuerySet:
    """The caller's queries of one batch (BLASTNA arrays) with the pointer / length arrays the C ABI takes,
    prepared once: a streaming caller sets the same queries up against several shards or options."""

    def __init__(self, queries):
        self.q = [np.ascontiguousarray(q, dtype=np.uint8) for q in queries]
        self.ptrs = (C.c_void_p * len(self.q))(*[q.ctypes.data for q in self.q])
        self.lens = (C.c_int32 * len(self.q))(*[len(q) for q in self.q])

    def __len__(self):
        return len(self.q)


class BlastPrelimSearch:
    """CBlastPrelimSearch analogue: one query batch against one resident shard."""

    def __init__(self, queries, options, seqsrc=None, upload=True, masks=None):
        """queries: list of BLASTNA arrays, or a QuerySet.  upload=False builds the host-side set-up only
        (no device needed).  masks: soft query masks [(query index, from, to)], inclusive plus-strand intervals."""
        L = lib()
        qs = queries if isinstance(queries, QuerySet) else QuerySet(queries)
        self._qs = qs
        self._q = qs.q
        ptrs, lens = qs.ptrs, qs.lens
        self.options = options
        self._b = C.c_void_p()
        masks = sorted(masks or [])
        n = len(masks)
        mq = (C.c_int32 * max(n, 1))(*[m[0] for m in masks])
        mf = (C.c_int32 * max(n, 1))(*[m[1] for m in masks])
        mt = (C.c_int32 * max(n, 1))(*[m[2] for m in masks])
        if upload and seqsrc is not None and seqsrc._h:
            _check(L.gbn_use_device(L.gbn_db_device(seqsrc._h)))     # the batch lives on the shard's GPU, whatever thread sets it up
        _check(L.gbn_batch_new_masked(C.byref(self._b), C.byref(options), len(self._q), ptrs, lens,
                                      n, mq, mf, mt, 1 if upload else 0))
        self._r = C.c_void_p()
        _check(L.gbn_results_new(C.byref(self._r)))
        self.seqsrc = seqsrc
        self.diagnostics = GbnDiagnostics()

    def info(self):
        L, b = lib(), self._b
        return dict(lut_type=L.gbn_batch_lut_type(b), lut_width=L.gbn_batch_lut_width(b),
                    scan_step=L.gbn_batch_scan_step(b), scan_path=L.gbn_batch_scan_path(b) if hasattr(L, "gbn_batch_scan_path") else -1, container=L.gbn_batch_diag_container(b),
                    gap_x_dropoff=L.gbn_batch_gap_x_dropoff(b))

    @property
    def contexts(self):
        n = lib().gbn_batch_num_contexts(self._b)
        p = lib().gbn_batch_contexts(self._b)
        return [p[i] for i in range(n)]

    def _grab(self, nf, pf, dt):
        n = nf(self._r)
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.frombuffer(C.string_at(pf(self._r), n * dt.itemsize), dtype=dt).copy()

    def run(self, seqsrc=None, keep_stages=False, clear=True):
        """Returns dict(hsps[, seeds, init_hits]); HSPs grouped by ascending oid."""
        L = lib()
        src = seqsrc or self.seqsrc
        if clear:
            L.gbn_results_clear(self._r)
        _check(L.gbn_prelim_search(self._b, src._h, self._r, C.byref(self.diagnostics),
                                   1 if keep_stages else 0, None, None))
        out = dict(hsps=self._grab(L.gbn_results_num_hsps, L.gbn_results_hsps, HSP_DT))
        if keep_stages:
            out["seeds"] = self._grab(L.gbn_results_num_seeds, L.gbn_results_seeds, SEED_DT)
            out["init_hits"] = self._grab(L.gbn_results_num_init_hits, L.gbn_results_init_hits, IHIT_DT)
        return out

    def begin(self, seqsrc=None):
        """Pipelined run: returns when only the gapped stage of this batch is still in flight."""
        L = lib()
        L.gbn_results_clear(self._r)
        _check(L.gbn_prelim_search_begin(self._b, (seqsrc or self.seqsrc)._h, self._r,
                                         C.byref(self.diagnostics), None, None))

    def end(self):
        """Wait for this batch's gapped stage if it is still in flight; returns the batch's HSPs."""
        L = lib()
        _check(L.gbn_prelim_search_end(self._r))
        return dict(hsps=self._grab(L.gbn_results_num_hsps, L.gbn_results_hsps, HSP_DT))

    def emit_lists(self):
        """The finished results as the HSP stream takes them: [(oid, HSP_DT records)] ascending (gbn_results_emit_lists)."""
        got = []

        def sink(arg, oid, ptr, n):
            got.append((int(oid), np.frombuffer(C.string_at(ptr, n * HSP_DT.itemsize), dtype=HSP_DT).copy()))
            return 0
        _check(lib().gbn_results_emit_lists(self._r, GbnHspListFn(sink), None))
        return got

    def scan_only(self, seqsrc=None, repeats=1):
        d = GbnDiagnostics()
        _check(lib().gbn_scan_only(self._b, (seqsrc or self.seqsrc)._h, repeats, C.byref(d)))
        return d

    def close(self):
        L = lib()
        if self._b:
            L.gbn_batch_free(self._b); self._b = None
        if self._r:
            L.gbn_results_free(self._r); self._r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BlastTracebackSearch:
    """CBlastTracebackSearch analogue (API/traceback_stage.cpp:198-306): final alignments of the lists a
    BlastHSPCollector kept for one query batch.  Host threads; the shard is read back from HBM."""

    def __init__(self, prelim, seqsrc=None):
        self._p = prelim
        self._src = seqsrc or prelim.seqsrc
        self._t = C.c_void_p()
        _check(lib().gbn_traceback_new(C.byref(self._t)))

    def run(self, hsps, list_starts, threads=0):
        """hsps, list_starts: BlastHSPCollector.close()[0:2].  -> (records TB_DT, ops list per record, query_starts)"""
        L = lib()
        h = np.ascontiguousarray(hsps, dtype=HSP_DT); st = np.ascontiguousarray(list_starts, dtype="<i8")
        _check(L.gbn_traceback_run(self._p._b, self._src._h, h.ctypes.data, st.ctypes.data, len(st) - 1, threads, self._t))
        return _read_traceback(self._t, len(self._p._q))

    def close(self):
        if self._t:
            lib().gbn_traceback_free(self._t); self._t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _read_traceback(t, nq):
    """(records TB_DT, ops per record, query_starts) of a GbnTraceback handle"""
    L = lib()
    n = L.gbn_traceback_num_hsps(t)
    qs = np.frombuffer(C.string_at(L.gbn_traceback_query_starts(t), (nq + 1) * 8), dtype="<i8").copy()
    if n == 0:
        return np.zeros(0, dtype=TB_DT), [], qs
    rec = np.frombuffer(C.string_at(L.gbn_traceback_hsps(t), n * TB_DT.itemsize), dtype=TB_DT).copy()
    nops = int(rec["ops_first"][-1] + rec["ops_count"][-1])
    op = np.frombuffer(C.string_at(L.gbn_traceback_ops(t), nops), dtype=np.uint8)
    ln = np.frombuffer(C.string_at(L.gbn_traceback_op_lengths(t), nops * 4), dtype="<i4")
    ops = [[(int(op[k]), int(ln[k])) for k in range(int(r["ops_first"]), int(r["ops_first"] + r["ops_count"]))] for r in rec]
    return rec, ops, qs


class SearchPipeline:
    """The host pipeline of gblastn_amd_host.hpp (CSearchPipeline) through its C ABI: query batches in,
    finished batches (traceback results) out in submission order; set-up, preliminary search and traceback
    run on their own host threads and overlap."""

    def __init__(self, options, seqsrc, trace_threads=2, traceback=True, overlap=True):
        self._p = C.c_void_p()
        self._src = seqsrc
        self._nq = {}
        _check(lib().gbn_pipeline_new(C.byref(self._p), C.byref(options), seqsrc._h, trace_threads, 1 if traceback else 0, 1 if overlap else 0))

    def submit(self, queries, masks=None):
        qs = queries if isinstance(queries, QuerySet) else QuerySet(queries)
        masks = sorted(masks or [])
        n = len(masks)
        mq = (C.c_int32 * max(n, 1))(*[m[0] for m in masks]); mf = (C.c_int32 * max(n, 1))(*[m[1] for m in masks])
        mt = (C.c_int32 * max(n, 1))(*[m[2] for m in masks])
        i = lib().gbn_pipeline_submit(self._p, len(qs), qs.ptrs, qs.lens, n, mq, mf, mt)
        if i < 0:
            raise BlastError(lib().gbn_last_error().decode())
        self._nq[i] = len(qs)
        return i

    def finish(self):
        lib().gbn_pipeline_finish(self._p)

    def next(self, read=True):
        """-> (batch number, (records, ops, query_starts) or None, diagnostics) or None when nothing is left"""
        L = lib()
        i, tb, col = C.c_int64(), C.c_void_p(), C.c_void_p()
        rc = L.gbn_pipeline_next(self._p, C.byref(i), C.byref(tb), C.byref(col))
        if rc == 1:
            return None
        _check(rc)
        d = GbnDiagnostics(); L.gbn_pipeline_diagnostics(self._p, C.byref(d))
        res = _read_traceback(tb, self._nq[i.value]) if (read and tb) else None
        return i.value, res, d

    def close(self):
        if self._p:
            lib().gbn_pipeline_free(self._p); self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BlastHSPCollector:
    """The HSP stream's collector writer (BlastHSPStreamWrite / Blast_HitListUpdate analogue):
    keeps, per query, the best min(2N, N+50) subjects.  Host only."""

    def __init__(self, num_queries, hitlist_size=500):
        self._c = C.c_void_p()
        _check(lib().gbn_collector_new(C.byref(self._c), num_queries, hitlist_size))

    def write(self, hsps):
        """hsps: HSP_DT records grouped by oid (what BlastPrelimSearch.run() returns)."""
        h = np.ascontiguousarray(hsps, dtype=HSP_DT)
        _check(lib().gbn_collector_write(self._c, h.ctypes.data, len(h)))

    def close(self):
        """-> (hsps, list_starts, list_queries): surviving lists in (oid, query) ascending order."""
        L = lib()
        _check(L.gbn_collector_close(self._c))
        nl, nh = L.gbn_collector_num_lists(self._c), L.gbn_collector_num_hsps(self._c)
        starts = np.frombuffer(C.string_at(L.gbn_collector_list_starts(self._c), (nl + 1) * 8), dtype="<i8").copy()
        queries = (np.frombuffer(C.string_at(L.gbn_collector_list_queries(self._c), nl * 4), dtype="<i4").copy()
                   if nl else np.zeros(0, dtype="<i4"))
        hsps = (np.frombuffer(C.string_at(L.gbn_collector_hsps(self._c), nh * HSP_DT.itemsize), dtype=HSP_DT).copy()
                if nh else np.zeros(0, dtype=HSP_DT))
        return hsps, starts, queries

    def free(self):
        if self._c:
            lib().gbn_collector_free(self._c); self._c = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class BlastDb:
    """BLAST database files (v4, nucleotide): alias / index / sequence files (CSeqDB analogue)."""

    def __init__(self, name):
        self._d = C.c_void_p()
        _check(lib().gbn_blastdb_open(C.byref(self._d), os.fsencode(name)))

    num_volumes = property(lambda self: lib().gbn_blastdb_num_volumes(self._d))
    num_seqs = property(lambda self: lib().gbn_blastdb_num_seqs(self._d))
    total_length = property(lambda self: lib().gbn_blastdb_total_length(self._d))
    max_length = property(lambda self: lib().gbn_blastdb_max_length(self._d))
    stat_num_seqs = property(lambda self: lib().gbn_blastdb_stat_num_seqs(self._d))
    stat_length = property(lambda self: lib().gbn_blastdb_stat_length(self._d))
    title = property(lambda self: lib().gbn_blastdb_title(self._d).decode())

    def volume_range(self, vol):
        a, n = C.c_int32(), C.c_int32()
        _check(lib().gbn_blastdb_volume_range(self._d, vol, C.byref(a), C.byref(n)))
        return a.value, n.value

    def seq_length(self, oid):
        return lib().gbn_blastdb_seq_length(self._d, oid)

    def ncbi2na(self, oid):
        n = self.seq_length(oid)
        if n < 0:
            raise BlastError("oid out of range")
        buf = np.zeros((n + 3) // 4, dtype=np.uint8)
        _check(lib().gbn_blastdb_get_ncbi2na(self._d, oid, buf.ctypes.data, len(buf)))
        return buf, n

    def blastna(self, oid, sentinels=False):
        n = self.seq_length(oid)
        if n < 0:
            raise BlastError("oid out of range")
        buf = np.zeros(n + (2 if sentinels else 0), dtype=np.uint8)
        _check(lib().gbn_blastdb_get_blastna(self._d, oid, buf.ctypes.data, len(buf), 1 if sentinels else 0))
        return buf

    def ambiguities(self, oid):
        k = lib().gbn_blastdb_num_ambiguities(self._d, oid)
        if k < 0:
            raise BlastError("oid out of range")
        st, ln, v = np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.uint8)
        if k:
            _check(lib().gbn_blastdb_get_ambiguities(self._d, oid, st.ctypes.data, ln.ctypes.data, v.ctypes.data, k))
        return st, ln, v

    def load_shard(self, first_oid=0, num_oids=None):
        """Subjects [first_oid, first_oid + num_oids) resident in HBM (needs a GPU) -> BlastSeqSrc."""
        h = C.c_void_p()
        n = self.num_seqs - first_oid if num_oids is None else num_oids
        _check(lib().gbn_blastdb_load_shard(self._d, first_oid, n, C.byref(h)))
        return BlastSeqSrc(h)

    def close(self):
        if self._d:
            lib().gbn_blastdb_close(self._d); self._d = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
