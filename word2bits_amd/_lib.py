"""ctypes binding of the C ABI in include/word2bits_hip.h, word2bits_corpus.h and word2bits_eval.h.

The shared library is built in-tree by word2bits_amd/csrc/Makefile (hipcc, gfx950).  Importing
this module fails loudly when it is missing: there is no Python/CPU fallback for the hot path.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("W2B_LIB", os.path.join(_HERE, "libword2bits_hip.so"))

W2B_OK, W2B_EINVAL, W2B_ENOGPU, W2B_EHIP, W2B_ENOMEM = 0, -1, -2, -3, -4
W2B_EUNSUPPORTED, W2B_ERCCL, W2B_ESTATE, W2B_EIO = -5, -6, -7, -8
ERROR_NAMES = {-1: "W2B_EINVAL", -2: "W2B_ENOGPU", -3: "W2B_EHIP", -4: "W2B_ENOMEM",
               -5: "W2B_EUNSUPPORTED", -6: "W2B_ERCCL", -7: "W2B_ESTATE", -8: "W2B_EIO"}


class W2bError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s (%d): %s" % (ERROR_NAMES.get(code, "?"), code, text))
        self.code = code


class Config(C.Structure):
    """struct w2b_config -- the reference's training globals (ref src/word2bits.cpp:45-61)."""
    _fields_ = [
        ("vocab_size", C.c_int64), ("train_words", C.c_int64), ("iter", C.c_int64),
        ("layer1_size", C.c_int32), ("window", C.c_int32), ("negative", C.c_int32),
        ("bitlevel", C.c_int32), ("num_threads", C.c_int32),
        ("alpha", C.c_float), ("sample", C.c_float), ("reg", C.c_float),
        ("compute_loss", C.c_int32), ("device", C.c_int32),
        ("worker_offset", C.c_int32), ("total_threads", C.c_int32), ("relaxed_coherence", C.c_int32),
        ("plain_worker_kernel", C.c_int32), ("exact_reduction", C.c_int32), ("reserved", C.c_int32 * 2),
    ]


class Tuning(C.Structure):
    """struct w2b_tuning -- the knobs that select code paths or trade fidelity for speed (include/word2bits_hip.h)."""
    _fields_ = [
        ("struct_size", C.c_int32), ("hot_rows_v", C.c_int32), ("hot_rows_u", C.c_int32), ("hot_period", C.c_int32),
        ("hot_cap", C.c_int32), ("force_row_desc", C.c_int32), ("grid_per_cu", C.c_int32), ("mem_mode", C.c_int32),
        ("atomic_rank", C.c_int32), ("atomic_cap", C.c_int32), ("hot_weight_permille", C.c_int32), ("window_refresh", C.c_int32),
        ("atomic_rank_u", C.c_int32), ("fresh_rank_u", C.c_int32), ("exchange_sat_updates", C.c_int32), ("refresh_rows_u", C.c_int32),
        ("exchange_rule", C.c_int32), ("exchange_tau_u", C.c_int32), ("exchange_tau_v", C.c_int32), ("concurrent_workers", C.c_int32),
    ]


class RowPlan(C.Structure):
    """struct w2b_row_plan -- what w2b_plan_rows decides for a launch (include/word2bits_hip.h)."""
    _fields_ = [("copies_u", C.c_int32), ("copies_v", C.c_int32), ("atomic_rank_u", C.c_int32), ("atomic_rank_v", C.c_int32),
                ("full_device", C.c_int32), ("merge_period", C.c_int32), ("row_group_kernel", C.c_int32), ("refresh_rows_u", C.c_int32),
                ("concurrent_workers", C.c_int32)]


vp, i32p, i64p, f32p, f64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), \
    C.POINTER(C.c_float), C.POINTER(C.c_double)
u64p = C.POINTER(C.c_uint64)

# name -> (restype, argtypes); every symbol declared in include/*.h
SIGNATURES = {
    "w2b_version": (C.c_char_p, []),
    "w2b_last_error": (C.c_char_p, []),
    "w2b_device_count": (C.c_int, []),
    "w2b_device_compute_units": (C.c_int, [C.c_int32]),
    "w2b_build_exp_table": (None, [f32p]),
    "w2b_build_unigram_table": (C.c_int, [i64p, C.c_int64, i32p, C.c_int64]),
    "w2b_build_keep_prob": (None, [i64p, C.c_int64, C.c_float, C.c_int64, f32p]),
    "w2b_quantize": (C.c_float, [C.c_float, C.c_int32]),
    "w2b_trainer_create": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
    "w2b_trainer_destroy": (None, [vp]),
    "w2b_plan_rows": (C.c_int, [C.POINTER(Config), C.POINTER(Tuning), i64p, C.c_int32, C.c_int32, C.POINTER(RowPlan)]),
    "w2b_get_tuning": (C.c_int, [vp, C.POINTER(Tuning)]),
    "w2b_set_tuning": (C.c_int, [vp, C.POINTER(Tuning)]),
    "w2b_init_net": (C.c_int, [vp]),
    "w2b_set_model": (C.c_int, [vp, f32p, f32p]),
    "w2b_get_model": (C.c_int, [vp, f32p, f32p]),
    "w2b_model_device_ptrs": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
    "w2b_export_quantized": (C.c_int, [vp, f32p]),
    "w2b_export_packed": (C.c_int, [vp, u64p]),
    "w2b_set_vocab_counts": (C.c_int, [vp, i64p, C.c_int64]),
    "w2b_set_unigram_table": (C.c_int, [vp, i32p, C.c_int64]),
    "w2b_set_exp_table": (C.c_int, [vp, f32p]),
    "w2b_set_corpus": (C.c_int, [vp, i32p, C.c_int64]),
    "w2b_set_corpus_device": (C.c_int, [vp, vp, C.c_int64]),
    "w2b_set_corpus_slice": (C.c_int, [vp, i32p, C.c_int64, C.c_int32]),
    "w2b_set_shards": (C.c_int, [vp, i64p, i32p]),
    "w2b_epoch_begin": (C.c_int, [vp]),
    "w2b_train_step": (C.c_int, [vp, C.c_int64]),
    "w2b_epoch_status": (C.c_int, [vp, i32p, i64p, f32p, f64p]),
    "w2b_epoch_poll": (C.c_int, [vp, C.c_int32, i32p, i64p, f32p, f64p]),
    "w2b_suggested_threads": (C.c_int, [vp, i32p]),
    "w2b_worker_kernel_info": (C.c_int, [vp, i32p, i32p, i32p, i32p, i32p]),
    "w2b_train_tuples": (C.c_int, [vp, C.c_int64, i32p, i32p, i32p, i32p, C.c_float, C.c_int32, f64p]),
    "w2b_train_tuples_device": (C.c_int, [vp, C.c_int64, vp, vp, vp, vp, C.c_float, C.c_int32]),
    "w2b_synchronize": (C.c_int, [vp]),
    "w2b_timing_enable": (C.c_int, [vp, C.c_int32]),
    "w2b_timing_read": (C.c_int, [vp, f64p, i64p]),
    "w2b_timing_launches": (C.c_int, [vp, f64p, C.c_int64, i64p]),
    "w2b_comm_unique_id": (C.c_int, [vp]),
    "w2b_comm_init": (C.c_int, [vp, C.c_int32, C.c_int32, vp]),
    "w2b_comm_count": (C.c_int, [vp, i32p]),
    "w2b_sync_replicas": (C.c_int, [vp, C.c_int32]),
    "w2b_suggested_exchange_words": (C.c_int64, [C.c_int64, C.c_int32]),
    "w2b_sync_stats": (C.c_int, [vp, i64p, f64p]),
    "w2b_exchange_init": (C.c_int, [vp]),
    "w2b_exchange_begin": (C.c_int, [vp, i64p, i64p]),
    "w2b_exchange_counts": (C.c_int, [vp, C.POINTER(vp), i64p]),
    "w2b_exchange_delta": (C.c_int, [vp, C.c_int64, C.POINTER(vp), i64p]),
    "w2b_exchange_apply": (C.c_int, [vp, C.c_int64, C.c_float]),
    "w2b_exchange_end": (C.c_int, [vp, C.c_int64]),
    # include/word2bits_corpus.h
    "w2b_corpus_load": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(vp)]),
    "w2b_corpus_load_ex": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(vp)]),
    "w2b_corpus_free": (None, [vp]),
    "w2b_corpus_vocab_size": (C.c_int64, [vp]),
    "w2b_corpus_train_words": (C.c_int64, [vp]),
    "w2b_corpus_file_size": (C.c_int64, [vp]),
    "w2b_corpus_word": (C.c_char_p, [vp, C.c_int64]),
    "w2b_corpus_counts": (i64p, [vp]),
    "w2b_corpus_search": (C.c_int32, [vp, C.c_char_p]),
    "w2b_corpus_num_tokens": (C.c_int64, [vp]),
    "w2b_corpus_tokens": (i32p, [vp]),
    "w2b_corpus_shards": (C.c_int, [vp, C.c_int32, i64p, i32p]),
    "w2b_packed_words_per_row": (C.c_int64, [C.c_int64, C.c_int32]),
    "w2b_pack_quantized": (C.c_int, [f32p, C.c_int64, C.c_int64, C.c_int32, u64p]),
    "w2b_unpack_quantized": (C.c_int, [u64p, C.c_int64, C.c_int64, C.c_int32, f32p]),
    "w2b_save_vectors_packed": (C.c_int, [C.c_char_p, vp, u64p, C.c_int64, C.c_int32]),
    "w2b_unpack_vectors_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32]),
    "w2b_save_vectors": (C.c_int, [C.c_char_p, vp, f32p, C.c_int64, C.c_int32]),
    # include/word2bits_eval.h
    "w2b_eval_load": (C.c_int, [C.c_char_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(vp)]),
    "w2b_eval_free": (None, [vp]),
    "w2b_eval_from_trainer": (C.c_int, [vp, C.c_int64, C.POINTER(C.c_char_p), C.c_int32, C.c_int64, C.c_int32, C.POINTER(vp)]),
    "w2b_eval_words": (C.c_int64, [vp]),
    "w2b_eval_size": (C.c_int64, [vp]),
    "w2b_eval_word": (C.c_char_p, [vp, C.c_int64]),
    "w2b_eval_lookup": (C.c_int64, [vp, C.c_char_p]),
    "w2b_eval_get_matrix": (C.c_int, [vp, f32p]),
    "w2b_eval_top1": (C.c_int, [vp, C.c_int64, i32p, i32p, i32p, i32p, f32p]),
    "w2b_eval_transcript": (C.c_int, [vp, C.c_char_p, C.c_int64, C.POINTER(vp), i64p]),
    "w2b_eval_free_text": (None, [vp]),
    "w2b_eval_set_kernel": (C.c_int, [vp, C.c_int32]),
    "w2b_eval_timing_read": (C.c_int, [vp, f64p, i64p, f64p]),
}

_lib = None


def lib():
    """Load libword2bits_hip.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "word2bits_amd: %s not found -- build it with `make -C word2bits_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()').  The HIP library is the "
            "product; there is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("W2B_LIB_ALLOW_MISSING") == "1" and not hasattr(L, name):
            # A/B runs against an OLDER build of the library (tools/gpu_session_*.sh); never the default, and never silent
            print("word2bits_amd: W2B_LIB_ALLOW_MISSING=1: symbol %s is missing from %s" % (name, LIB_PATH), file=sys.stderr)
            continue
        f = getattr(L, name)        # AttributeError here = header/library mismatch
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise W2bError(rc, lib().w2b_last_error().decode("utf-8", "replace"))
