"""ctypes binding of include/bpp.h.  There is no fallback: a missing or unloadable libbpp_hip.so raises."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_size_t, c_uint8, c_uint32, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbpp_hip.so")
# BPP_LIB_PATH: load ANOTHER build of the library instead (measurement builds such as -DBPP_KP_PHASES: tools/gpu_kp_phases.sh).  The
# product's own .so is never swapped in place for one; whoever sets the variable names the file and gets exactly that file.
if os.environ.get("BPP_LIB_PATH"):
    LIB_PATH = os.path.abspath(os.environ["BPP_LIB_PATH"])

u8p = POINTER(c_uint8)


class VerifyItem(Structure):
    _fields_ = [("proof", c_void_p), ("proof_len", c_size_t), ("commitments32", c_void_p), ("m", c_uint32),
                ("min_values", c_void_p), ("min_present", c_void_p), ("seed_nonce32", c_void_p),
                ("transcript_state", c_void_p), ("transcript_label", c_void_p), ("label_len", c_size_t)]


class ProveItem(Structure):
    _fields_ = [("values", c_void_p), ("blindings32", c_void_p), ("commitments32", c_void_p), ("m", c_uint32),
                ("min_values", c_void_p), ("min_present", c_void_p), ("seed_nonce32", c_void_p),
                ("transcript_state", c_void_p), ("transcript_label", c_void_p), ("label_len", c_size_t),
                ("rng_bytes", c_void_p), ("rng_len", c_size_t)]


class PackedBatch(Structure):
    """bpp_packed_batch: a homogeneous batch as contiguous arrays"""
    _fields_ = [("n_items", c_size_t), ("proofs", c_void_p), ("proof_len", c_size_t), ("proof_stride", c_size_t),
                ("commitments32", c_void_p), ("m", c_uint32), ("min_values", c_void_p), ("min_present", c_void_p),
                ("seed_nonces32", c_void_p), ("seed_present", c_void_p), ("transcript_state", c_void_p),
                ("transcript_label", c_void_p), ("label_len", c_size_t)]


class ShardResult(Structure):
    """bpp_shard_result: outcome of one batch of a sharded wave"""
    _fields_ = [("code", c_int), ("tier", c_int), ("rank", c_int), ("index", c_uint32), ("msg", ctypes.c_char * 160)]


class ShardTiming(Structure):
    _fields_ = [(n, c_float) for n in ("enqueue1_ms", "wait1_ms", "gather1_ms", "chains_ms", "enqueue2_ms", "wait2_ms",
                                       "gather2_ms")] + [("batches", c_uint32)]


class Profile(Structure):
    _fields_ = [(n, c_float) for n in ("transcripts_ms", "decompress_ms", "chain_host_ms", "scalars_ms", "reduce_ms",
                                       "msm_digits_ms", "msm_sort_ms", "msm_accumulate_ms", "msm_bucket_reduce_ms",
                                       "msm_final_ms", "total_ms")] + \
               [(n, c_uint32) for n in ("msm_terms", "msm_window_bits", "msm_windows", "msm_groups")] + [("masks_ms", c_float), ("chain_device_ms", c_float)]


class ProveProfile(Structure):
    _fields_ = [("fb_msm_ms", c_float), ("total_ms", c_float), ("fb_terms", c_uint64), ("fb_launches", c_uint32),
                ("fb_window_bits", c_uint32), ("fb_windows", c_uint32), ("sub_batches", c_uint32)]


class RuntimeInfo(Structure):
    """bpp_runtime_info: what the library sees of its runtime preconditions (hardware queues, contexts, the small-call gate)"""
    _fields_ = [("device", c_int), ("contexts", c_uint32), ("contexts_peak", c_uint32), ("hw_queues", c_uint32),
                ("host_threads", c_uint32), ("small_call_limit", c_uint32), ("small_calls_in_flight", c_uint32),
                ("small_calls", c_uint64), ("small_calls_queued", c_uint64), ("oversubscribed", c_uint32)]


# bpp_all_gather_fn: int (*)(void *user, const void *send, void *recv, size_t bytes_per_rank)
ALL_GATHER_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_size_t)

# every symbol include/bpp.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("bpp_ctx_create", c_int, [POINTER(c_void_p), c_int]),
    ("bpp_ctx_create_on_stream", c_int, [POINTER(c_void_p), c_int, c_void_p]),
    ("bpp_ctx_destroy", None, [c_void_p]),
    ("bpp_ctx_last_error", c_char_p, [c_void_p]),
    ("bpp_ctx_set_option", c_int, [c_void_p, c_char_p, c_int]),
    ("bpp_precomp_create", c_int, [c_void_p, c_void_p, c_size_t, POINTER(c_uint64)]),
    ("bpp_precomp_destroy", c_int, [c_void_p, c_uint64]),
    ("bpp_precomp_retain", c_int, [c_void_p, c_uint64]),
    ("bpp_msm_mixed", c_int, [c_void_p, c_uint64, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("bpp_msm_vartime", c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("bpp_msm_vartime_batched", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("bpp_params_create", c_int, [c_void_p, c_uint32, c_uint32, c_uint32, c_void_p, c_void_p, POINTER(c_uint64)]),
    ("bpp_params_destroy", c_int, [c_void_p, c_uint64]),
    ("bpp_params_retain", c_int, [c_void_p, c_uint64]),
    ("bpp_params_export", c_int, [c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("bpp_pedersen_commit", c_int, [c_void_p, c_uint64, c_void_p, c_void_p, c_uint32, c_size_t, c_void_p]),
    ("bpp_verify_batch", c_int, [c_void_p, c_uint64, POINTER(VerifyItem), c_size_t, c_int, c_size_t, c_void_p,
                                 c_void_p, c_void_p, c_size_t]),
    ("bpp_verify_batch_with_challenges", c_int, [c_void_p, c_uint64, POINTER(VerifyItem), c_size_t, POINTER(c_void_p),
                                                 c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t]),
    ("bpp_batch_upload", c_int, [c_void_p, c_uint64, POINTER(VerifyItem), c_size_t, POINTER(c_uint64), c_void_p,
                                 c_size_t]),
    ("bpp_batch_upload_packed", c_int, [c_void_p, c_uint64, POINTER(PackedBatch), POINTER(c_uint64), c_void_p, c_size_t]),
    ("bpp_verify_batch_packed", c_int, [c_void_p, c_uint64, POINTER(PackedBatch), c_int, c_size_t, c_void_p, c_void_p,
                                        c_void_p, c_size_t]),
    ("bpp_ctx_pipeline_depth", c_int, [c_void_p, c_uint32]),
    ("bpp_verify_submit_packed", c_int, [c_void_p, c_uint64, POINTER(PackedBatch), c_int, c_size_t, POINTER(c_uint64),
                                         c_void_p, c_size_t]),
    ("bpp_verify_collect", c_int, [c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_size_t]),
    ("bpp_batch_destroy", c_int, [c_void_p, c_uint64]),
    ("bpp_verify_resident", c_int, [c_void_p, c_uint64, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t]),
    ("bpp_verify_phase1", c_int, [c_void_p, c_uint64, c_void_p, c_void_p, c_size_t]),
    ("bpp_weights_from_chain", c_int, [c_void_p, c_size_t, c_void_p]),
    ("bpp_weights_from_chains", c_int, [c_void_p, c_size_t, c_size_t, c_void_p]),
    ("bpp_verify_phase2", c_int, [c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_size_t]),
    ("bpp_accumulators_sum_is_identity", c_int, [c_void_p, c_void_p, c_size_t, POINTER(c_int)]),
    ("bpp_comm_unique_id", c_int, [c_void_p]),
    ("bpp_comm_create", c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_void_p)]),
    ("bpp_comm_adopt", c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_void_p)]),
    ("bpp_comm_create_callbacks", c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, POINTER(c_void_p)]),
    ("bpp_comm_create_local", c_int, [c_void_p, c_uint64, c_int, c_int, POINTER(c_void_p)]),
    ("bpp_comm_destroy", None, [c_void_p]),
    ("bpp_comm_last_error", c_char_p, [c_void_p]),
    ("bpp_comm_set_timeout", c_int, [c_void_p, c_uint32]),
    ("bpp_comm_last_timing", c_int, [c_void_p, POINTER(ShardTiming)]),
    ("bpp_verify_sharded", c_int, [c_void_p, c_void_p, c_uint64, POINTER(c_uint32), POINTER(c_int), POINTER(c_int), c_void_p,
                                   c_size_t]),
    ("bpp_verify_sharded_wave", c_int, [c_void_p, POINTER(c_void_p), POINTER(c_uint64), c_size_t, POINTER(c_uint32),
                                        POINTER(ShardResult)]),
    ("bpp_verify_sharded_groups", c_int, [c_void_p, c_void_p, c_uint64, c_size_t, POINTER(c_uint32), POINTER(ShardResult)]),
    ("bpp_verify_sharded_groups_wave", c_int, [c_void_p, POINTER(c_void_p), POINTER(c_uint64), c_size_t, c_size_t, POINTER(c_uint32),
                                               POINTER(ShardResult)]),
    ("bpp_verify_resident_groups", c_int, [c_void_p, c_uint64, POINTER(c_uint32), c_size_t, POINTER(ShardResult)]),
    ("bpp_verify_resident_groups_actions", c_int, [c_void_p, c_uint64, POINTER(c_uint32), c_size_t, POINTER(c_int), POINTER(ShardResult),
                                                   c_void_p, c_void_p]),
    ("bpp_batcher_verify_action", c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_char_p, c_size_t]),
    ("bpp_batcher_set_limits", c_int, [c_void_p, c_uint32, c_uint32]),
    ("bpp_batcher_largest_pool", c_int, [c_void_p, POINTER(c_uint32), POINTER(c_uint32)]),
    ("bpp_runtime_info_get", c_int, [c_void_p, POINTER(RuntimeInfo)]),
    ("bpp_small_call_limit", c_int, [c_void_p, c_int]),
    ("bpp_batcher_create", c_int, [c_void_p, c_uint64, c_void_p, c_uint32, c_uint32, c_uint32, POINTER(c_void_p)]),
    ("bpp_batcher_verify", c_int, [c_void_p, c_void_p, c_char_p, c_size_t]),
    ("bpp_batcher_stats", c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
    ("bpp_batcher_destroy", None, [c_void_p]),
    ("bpp_shard_local_trailer", c_int, [c_void_p, c_void_p, c_void_p, c_uint32, c_uint32, c_void_p]),
    ("bpp_shard_trailer", c_int, [c_int, c_int, c_uint32, c_char_p, c_void_p]),
    ("bpp_shard_resolve", c_int, [c_void_p, c_size_t, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_uint32), c_void_p,
                                  c_size_t]),
    ("bpp_prove_batch", c_int, [c_void_p, c_uint64, POINTER(ProveItem), c_size_t, c_void_p, c_size_t, POINTER(c_size_t),
                                c_void_p, c_size_t]),
    ("bpp_batch_trace", c_int, [c_void_p, c_uint64, c_int, c_void_p, c_size_t, POINTER(c_size_t)]),
    ("bpp_batch_shape", c_int, [c_void_p, c_uint64, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32),
                                POINTER(c_uint32), POINTER(c_uint32)]),
    ("bpp_profile_enable", c_int, [c_void_p, c_int]),
    ("bpp_profile_get", c_int, [c_void_p, POINTER(Profile)]),
    ("bpp_prove_profile_get", c_int, [c_void_p, POINTER(ProveProfile)]),
    ("bpp_batch_prepare", c_int, [c_void_p, c_uint64, c_size_t]),
    ("bpp_host_threads", c_int, []),
    ("bpp_host_pool_cpu_ns", c_uint64, []),
    ("bpp_device_chain_stats", c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
    ("bpp_shader_clock", c_int, [c_void_p, c_uint32, POINTER(c_double)]),
    ("bpp_transcript_new", c_int, [c_void_p, c_size_t, c_void_p]),
    ("bpp_batch_secret_bytes", c_int, [c_void_p, c_uint64, POINTER(c_uint64)]),
    ("bpp_prove_secret_bytes", c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
]

_lib = None


def load():
    """Load libbpp_hip.so and bind every declared symbol; raises if the extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libbpp_hip.so is missing (%s): run __graft_entry__.build(); there is no CPU fallback"
                           % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
