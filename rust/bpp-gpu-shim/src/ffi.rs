//! Raw declarations: one `extern "C"` item per symbol of include/bpp.h, same order, same argument meaning.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct bpp_ctx {
    _p: [u8; 0],
}

pub const BPP_OK: c_int = 0;
pub const BPP_ERR_VERIFICATION_FAILED: c_int = 1;
pub const BPP_ERR_INVALID_ARGUMENT: c_int = 2;
pub const BPP_ERR_INVALID_LENGTH: c_int = 3;
pub const BPP_ERR_INVALID_BLAKE2B: c_int = 4;
pub const BPP_ERR_SIZE_OVERFLOW: c_int = 5;
pub const BPP_ERR_ENGINE: c_int = -1;
pub const BPP_ERR_NO_DEVICE: c_int = -2;
pub const BPP_ERR_BAD_HANDLE: c_int = -3;
pub const BPP_ERR_COMM: c_int = -4;
// where inside RangeProof::verify a check failed (src/range_proof.rs:756-1065): orders findings across shards
pub const BPP_TIER_NONE: c_int = 0;
pub const BPP_TIER_CONSTRUCTION: c_int = 1;
pub const BPP_TIER_DEGREE: c_int = 2;
pub const BPP_TIER_PROMISE: c_int = 3;
pub const BPP_TIER_STATEMENT_POINT: c_int = 4;
pub const BPP_TIER_PASS1: c_int = 5;
pub const BPP_TIER_PASS2: c_int = 6;
pub const BPP_TIER_MSM: c_int = 7;
pub const BPP_TIER_ENGINE: c_int = 255;
pub const BPP_SHARD_TRAILER_BYTES: usize = 128;
pub const BPP_REFERENCE_CHUNK: usize = 256;

/// bpp_verify_item: one (transcript, statement, proof) triple of `RangeProof::verify_batch` (src/range_proof.rs:712-717)
#[repr(C)]
pub struct bpp_verify_item {
    pub proof: *const u8,
    pub proof_len: usize,
    pub commitments32: *const u8,
    pub m: u32,
    pub min_values: *const u64,
    pub min_present: *const u8,
    pub seed_nonce32: *const u8,
    pub transcript_state: *const u8,
    pub transcript_label: *const u8,
    pub label_len: usize,
}

/// bpp_prove_item: one (transcript, statement, witness, rng) quadruple of `RangeProof::prove_with_rng` (:232-237)
#[repr(C)]
pub struct bpp_prove_item {
    pub values: *const u64,
    pub blindings32: *const u8,
    pub commitments32: *const u8,
    pub m: u32,
    pub min_values: *const u64,
    pub min_present: *const u8,
    pub seed_nonce32: *const u8,
    pub transcript_state: *const u8,
    pub transcript_label: *const u8,
    pub label_len: usize,
    pub rng_bytes: *const u8,
    pub rng_len: usize,
}

/// bpp_packed_batch: a homogeneous batch (`&[RangeStatement]`, `&[RangeProof]` of one shape) as contiguous arrays
#[repr(C)]
pub struct bpp_packed_batch {
    pub n_items: usize,
    pub proofs: *const u8,
    pub proof_len: usize,
    pub proof_stride: usize,
    pub commitments32: *const u8,
    pub m: u32,
    pub min_values: *const u64,
    pub min_present: *const u8,
    pub seed_nonces32: *const u8,
    pub seed_present: *const u8,
    pub transcript_state: *const u8,
    pub transcript_label: *const u8,
    pub label_len: usize,
}

#[repr(C)]
pub struct bpp_comm {
    _p: [u8; 0],
}
#[repr(C)]
pub struct bpp_batcher {
    _p: [u8; 0],
}

/// outcome of one batch of a sharded wave (bpp_verify_sharded_wave)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct bpp_shard_result {
    pub code: c_int,
    pub tier: c_int,
    pub rank: c_int,
    pub index: u32,
    pub msg: [c_char; 160],
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct bpp_shard_timing {
    pub enqueue1_ms: f32,
    pub wait1_ms: f32,
    pub gather1_ms: f32,
    pub chains_ms: f32,
    pub enqueue2_ms: f32,
    pub wait2_ms: f32,
    pub gather2_ms: f32,
    pub batches: u32,
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct bpp_profile {
    pub transcripts_ms: f32,
    pub decompress_ms: f32,
    pub chain_host_ms: f32,
    pub scalars_ms: f32,
    pub reduce_ms: f32,
    pub msm_digits_ms: f32,
    pub msm_sort_ms: f32,
    pub msm_accumulate_ms: f32,
    pub msm_bucket_reduce_ms: f32,
    pub msm_final_ms: f32,
    pub total_ms: f32,
    pub msm_terms: u32,
    pub msm_window_bits: u32,
    pub msm_windows: u32,
    pub msm_groups: u32,
    pub masks_ms: f32,
    pub chain_device_ms: f32,
}

/// bpp_runtime_info: what the library sees of its runtime preconditions ("Runtime preconditions")
#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct bpp_runtime_info {
    pub device: c_int,
    pub contexts: u32,
    pub contexts_peak: u32,
    pub hw_queues: u32,
    pub host_threads: u32,
    pub small_call_limit: u32,
    pub small_calls_in_flight: u32,
    pub small_calls: u64,
    pub small_calls_queued: u64,
    pub oversubscribed: u32,
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct bpp_prove_profile {
    pub fb_msm_ms: f32,
    pub total_ms: f32,
    pub fb_terms: u64,
    pub fb_launches: u32,
    pub fb_window_bits: u32,
    pub fb_windows: u32,
    pub sub_batches: u32,
}

/// `int (*bpp_all_gather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank)` (include/bpp.h)
pub type bpp_all_gather_fn = Option<unsafe extern "C" fn(user: *mut c_void, send: *const c_void, recv: *mut c_void, bytes_per_rank: usize) -> c_int>;

extern "C" {
    pub fn bpp_ctx_create(out: *mut *mut bpp_ctx, device_id: c_int) -> c_int;
    pub fn bpp_ctx_create_on_stream(out: *mut *mut bpp_ctx, device_id: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn bpp_ctx_destroy(ctx: *mut bpp_ctx);
    pub fn bpp_ctx_last_error(ctx: *mut bpp_ctx) -> *const c_char;
    pub fn bpp_ctx_set_option(ctx: *mut bpp_ctx, name: *const c_char, value: c_int) -> c_int;
    // runtime preconditions, the admission gate for small calls
    pub fn bpp_runtime_info_get(ctx: *mut bpp_ctx, out: *mut bpp_runtime_info) -> c_int;
    pub fn bpp_small_call_limit(ctx: *mut bpp_ctx, limit: c_int) -> c_int;
    // B1: VartimePrecomputedMultiscalarMul / VartimeMultiscalarMul / MultiscalarMul (src/traits.rs:40-43, src/ristretto.rs:28-64)
    pub fn bpp_precomp_create(ctx: *mut bpp_ctx, points32: *const u8, count: usize, handle: *mut u64) -> c_int;
    pub fn bpp_precomp_destroy(ctx: *mut bpp_ctx, handle: u64) -> c_int;
    pub fn bpp_precomp_retain(ctx: *mut bpp_ctx, handle: u64) -> c_int;
    pub fn bpp_msm_mixed(ctx: *mut bpp_ctx, handle: u64, static_scalars32: *const u8, n_static: usize, dyn_scalars32: *const u8,
                         dyn_points32: *const u8, n_dyn: usize, out_point32: *mut u8) -> c_int;
    pub fn bpp_msm_vartime(ctx: *mut bpp_ctx, scalars32: *const u8, points32: *const u8, n: usize, out_point32: *mut u8) -> c_int;
    pub fn bpp_msm_vartime_batched(ctx: *mut bpp_ctx, scalars32: *const u8, points32: *const u8, group_off: *const u32,
                                   n_groups: usize, out_points32: *mut u8) -> c_int;
    // B2: RangeParameters::init (src/range_parameters.rs:32-58), PedersenGens::commit (src/generators/pedersen_gens.rs:112-122)
    pub fn bpp_params_create(ctx: *mut bpp_ctx, bit_length: u32, max_aggregation: u32, extension_degree: u32, h_base32: *const u8,
                             g_bases32: *const u8, params: *mut u64) -> c_int;
    pub fn bpp_params_destroy(ctx: *mut bpp_ctx, params: u64) -> c_int;
    pub fn bpp_params_retain(ctx: *mut bpp_ctx, params: u64) -> c_int;
    pub fn bpp_params_export(ctx: *mut bpp_ctx, params: u64, gi_out32: *mut u8, hi_out32: *mut u8, h_out32: *mut u8, g_out32: *mut u8) -> c_int;
    pub fn bpp_pedersen_commit(ctx: *mut bpp_ctx, params: u64, values: *const u64, blindings32: *const u8, n_blind: u32, count: usize,
                               commitments32: *mut u8) -> c_int;
    // B2: RangeProof::verify_batch / verify (src/range_proof.rs:712-1065)
    pub fn bpp_verify_batch(ctx: *mut bpp_ctx, params: u64, items: *const bpp_verify_item, n_items: usize, action: c_int, chunk: usize,
                            masks_out: *mut u8, mask_present: *mut u8, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn bpp_verify_batch_with_challenges(ctx: *mut bpp_ctx, params: u64, items: *const bpp_verify_item, n_items: usize,
                                            challenges32: *const *const u8, rng_out32: *const u8, action: c_int, chunk: usize,
                                            masks_out: *mut u8, mask_present: *mut u8, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn bpp_batch_upload(ctx: *mut bpp_ctx, params: u64, items: *const bpp_verify_item, n_items: usize, batch: *mut u64,
                            errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn bpp_batch_destroy(ctx: *mut bpp_ctx, batch: u64) -> c_int;
    pub fn bpp_batch_prepare(ctx: *mut bpp_ctx, batch: u64, chunk: usize) -> c_int;
    pub fn bpp_verify_resident(ctx: *mut bpp_ctx, batch: u64, action: c_int, chunk: usize, masks_out: *mut u8, mask_present: *mut u8,
                               errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    // packed form + pipelined host-buffers-in form (upload k+1 overlaps verify k inside one context)
    pub fn bpp_batch_upload_packed(ctx: *mut bpp_ctx, params: u64, input: *const bpp_packed_batch, batch: *mut u64, errbuf: *mut c_char,
                                   errbuf_len: usize) -> c_int;
    pub fn bpp_verify_batch_packed(ctx: *mut bpp_ctx, params: u64, input: *const bpp_packed_batch, action: c_int, chunk: usize,
                                   masks_out: *mut u8, mask_present: *mut u8, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn bpp_ctx_pipeline_depth(ctx: *mut bpp_ctx, depth: u32) -> c_int;
    pub fn bpp_verify_submit_packed(ctx: *mut bpp_ctx, params: u64, input: *const bpp_packed_batch, action: c_int, chunk: usize,
                                    ticket: *mut u64, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn bpp_verify_collect(ctx: *mut bpp_ctx, ticket: u64, masks_out: *mut u8, mask_present: *mut u8, errbuf: *mut c_char,
                              errbuf_len: usize) -> c_int;
    // ONE reference batch sharded over the GPUs of a node: RCCL all_gathers on device buffers inside the library
    pub fn bpp_comm_unique_id(id128: *mut u8) -> c_int;
    pub fn bpp_comm_create(ctx: *mut bpp_ctx, id128: *const u8, rank: c_int, world: c_int, out: *mut *mut bpp_comm) -> c_int;
    pub fn bpp_comm_adopt(ctx: *mut bpp_ctx, nccl_comm: *mut c_void, rank: c_int, world: c_int, out: *mut *mut bpp_comm) -> c_int;
    /// The caller's own transport (MPI / TCP / gloo): `all_gather` sees HOST memory, blocks, returns 0 or an error (include/bpp.h).
    pub fn bpp_comm_create_callbacks(ctx: *mut bpp_ctx, rank: c_int, world: c_int, all_gather: bpp_all_gather_fn, user: *mut c_void,
                                     out: *mut *mut bpp_comm) -> c_int;
    pub fn bpp_comm_create_local(ctx: *mut bpp_ctx, group_id: u64, rank: c_int, world: c_int, out: *mut *mut bpp_comm) -> c_int;
    pub fn bpp_comm_destroy(comm: *mut bpp_comm);
    pub fn bpp_comm_last_error(comm: *mut bpp_comm) -> *const c_char;
    pub fn bpp_comm_set_timeout(comm: *mut bpp_comm, timeout_ms: u32) -> c_int;
    pub fn bpp_comm_last_timing(comm: *mut bpp_comm, out: *mut bpp_shard_timing) -> c_int;
    pub fn bpp_verify_sharded(comm: *mut bpp_comm, ctx: *mut bpp_ctx, batch: u64, counts: *const u32, tier_out: *mut c_int,
                              rank_out: *mut c_int, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn bpp_verify_sharded_wave(comm: *mut bpp_comm, ctxs: *const *mut bpp_ctx, batches: *const u64, k: usize, counts: *const u32,
                                   results: *mut bpp_shard_result) -> c_int;
    pub fn bpp_verify_sharded_groups(comm: *mut bpp_comm, ctx: *mut bpp_ctx, batch: u64, n_groups: usize, counts: *const u32,
                                     results: *mut bpp_shard_result) -> c_int;
    pub fn bpp_verify_sharded_groups_wave(comm: *mut bpp_comm, ctxs: *const *mut bpp_ctx, batches: *const u64, k: usize, n_groups: usize,
                                          counts: *const u32, results: *mut bpp_shard_result) -> c_int;
    // reference batches of different sizes as the groups of one call; the pool of many callers' small calls
    pub fn bpp_verify_resident_groups(ctx: *mut bpp_ctx, batch: u64, group_first: *const u32, n_groups: usize, results: *mut bpp_shard_result) -> c_int;
    pub fn bpp_batcher_create(ctx: *mut bpp_ctx, params: u64, shape: *const bpp_packed_batch, lanes: u32, max_wait_us: u32, max_calls: u32,
                              out: *mut *mut bpp_batcher) -> c_int;
    pub fn bpp_batcher_verify(b: *mut bpp_batcher, input: *const bpp_packed_batch, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn bpp_verify_resident_groups_actions(ctx: *mut bpp_ctx, batch: u64, group_first: *const u32, n_groups: usize, actions: *const c_int,
                                              results: *mut bpp_shard_result, masks_out: *mut u8, mask_present: *mut u8) -> c_int;
    pub fn bpp_batcher_verify_action(b: *mut bpp_batcher, input: *const bpp_packed_batch, action: c_int, masks_out: *mut u8,
                                     mask_present: *mut u8, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn bpp_batcher_set_limits(b: *mut bpp_batcher, max_calls: u32, max_proofs: u32) -> c_int;
    pub fn bpp_batcher_largest_pool(b: *mut bpp_batcher, calls: *mut u32, proofs: *mut u32) -> c_int;
    pub fn bpp_batcher_stats(b: *mut bpp_batcher, pooled_calls: *mut u64, engine_calls: *mut u64, solo_calls: *mut u64) -> c_int;
    pub fn bpp_batcher_destroy(b: *mut bpp_batcher);
    pub fn bpp_shard_local_trailer(defer: *const u8, status: *const u32, rounds_bad: *const u8, n: u32, first_index: u32,
                                   trailer_out: *mut u8) -> c_int;
    pub fn bpp_shard_trailer(tier: c_int, code: c_int, index: u32, msg: *const c_char, trailer_out: *mut u8) -> c_int;
    pub fn bpp_shard_resolve(trailers: *const u8, stride: usize, world: c_int, tier_out: *mut c_int, rank_out: *mut c_int,
                             index_out: *mut u32, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    // phased form of the same (callers that bring their own transport)
    pub fn bpp_verify_phase1(ctx: *mut bpp_ctx, batch: u64, rng_out32: *mut u8, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn bpp_weights_from_chain(rng32_all: *const u8, n_total: usize, weights32_out: *mut u8) -> c_int;
    pub fn bpp_weights_from_chains(rng32_all: *const u8, n_groups: usize, n_per_group: usize, weights32_out: *mut u8) -> c_int;
    pub fn bpp_verify_phase2(ctx: *mut bpp_ctx, batch: u64, weights32: *const u8, accumulator128: *mut u8, errbuf: *mut c_char,
                             errbuf_len: usize) -> c_int;
    pub fn bpp_accumulators_sum_is_identity(ctx: *mut bpp_ctx, accumulators128: *const u8, n: usize, is_identity: *mut c_int) -> c_int;
    // B2: RangeProof::prove_with_rng (src/range_proof.rs:232-608)
    pub fn bpp_prove_batch(ctx: *mut bpp_ctx, params: u64, items: *const bpp_prove_item, n_items: usize, proofs_out: *mut u8,
                           proof_stride: usize, proof_len: *mut usize, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    // diagnostics
    pub fn bpp_batch_trace(ctx: *mut bpp_ctx, batch: u64, what: c_int, out: *mut u8, out_len: usize, written: *mut usize) -> c_int;
    pub fn bpp_batch_shape(ctx: *mut bpp_ctx, batch: u64, n_items: *mut u32, max_rounds: *mut u32, max_mn: *mut u32, total_dyn: *mut u32,
                           groups: *mut u32) -> c_int;
    pub fn bpp_profile_enable(ctx: *mut bpp_ctx, on: c_int) -> c_int;
    pub fn bpp_profile_get(ctx: *mut bpp_ctx, out: *mut bpp_profile) -> c_int;
    pub fn bpp_prove_profile_get(ctx: *mut bpp_ctx, out: *mut bpp_prove_profile) -> c_int;
    pub fn bpp_host_threads() -> c_int;
    pub fn bpp_host_pool_cpu_ns() -> u64;
    pub fn bpp_device_chain_stats(ctx: *mut bpp_ctx, calls: *mut u64, redraws: *mut u64) -> c_int;
    pub fn bpp_transcript_new(label: *const u8, label_len: usize, state203: *mut u8) -> c_int;
    pub fn bpp_batch_secret_bytes(ctx: *mut bpp_ctx, batch: u64, nonzero: *mut u64) -> c_int;
    pub fn bpp_prove_secret_bytes(ctx: *mut bpp_ctx, examined: *mut u64, nonzero: *mut u64) -> c_int;
    pub fn bpp_shader_clock(ctx: *mut bpp_ctx, window_us: u32, ghz: *mut f64) -> c_int;
}
