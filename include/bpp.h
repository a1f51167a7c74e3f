/* bpp.h -- C ABI of libbpp_hip.so, the MI355X (gfx950) engine for the Bulletproofs+ range-proof hot path.
 *
 * Drop-in boundary for tari_bulletproofs_plus 0.4.1 (reference = /root/reference, Rust).  Two seams:
 *
 *  B1 (trait level)   the three curve25519-dalek multiscalar traits the reference is generic over
 *                     (src/traits.rs:40-43, src/protocols/curve_point_protocol.rs:18-36, impl src/ristretto.rs:28-64)
 *  B2 (protocol level) the bodies of RangeProof::verify (src/range_proof.rs:756-1065, entered through
 *                     verify_batch :712-752) and RangeParameters::init (src/range_parameters.rs:32-58).
 *
 * Conventions
 *  - points cross the boundary as 32-byte canonical ristretto255 encodings, scalars as 32-byte canonical
 *    little-endian integers mod l; no C++ or torch types; caller owns every buffer; nothing is retained.
 *  - return value: 0 = Ok; 1..5 = the reference's ProofError variants (src/errors.rs:11-28); negative = engine fault.
 *  - a ctx is bound to one HIP device and one stream and serialises its own calls; batch handles belong to their ctx.
 *  - params / precomp handles are process-wide, reference-counted, read-only objects (the reference shares its
 *    generators and `Precomputation: Send + Sync` tables through Arc: src/traits.rs:42,
 *    src/generators/bulletproof_gens.rs:52,103): ANY ctx of the same device may pass a live handle, from any thread,
 *    concurrently; bpp_*_retain is Arc::clone, bpp_*_destroy is drop.  Device tables are freed when the last
 *    reference, resident batch and call in flight is gone.
 *  - every entry point fails (negative code) if no gfx950 device is usable: there is NO CPU fallback.
 */
#ifndef BPP_H
#define BPP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ProofError mapping (src/errors.rs:11-28) */
#define BPP_OK 0
#define BPP_ERR_VERIFICATION_FAILED 1
#define BPP_ERR_INVALID_ARGUMENT 2
#define BPP_ERR_INVALID_LENGTH 3
#define BPP_ERR_INVALID_BLAKE2B 4
#define BPP_ERR_SIZE_OVERFLOW 5
/* engine faults (no reference analogue) */
#define BPP_ERR_ENGINE (-1)    /* HIP runtime error, see bpp_ctx_last_error */
#define BPP_ERR_NO_DEVICE (-2) /* no usable gfx950 device */
#define BPP_ERR_BAD_HANDLE (-3)
#define BPP_ERR_COMM (-4)      /* RCCL failure (library missing, communicator error): sharded entry points only */

/* Where inside RangeProof::verify (src/range_proof.rs:756-1065) a check failed.  Lower = earlier in the reference's order
 * of checks; shards of one reference batch combine their findings by (tier, rank) -- see bpp_verify_sharded. */
#define BPP_TIER_NONE 0
#define BPP_TIER_CONSTRUCTION 1     /* RangeStatement::init / RangeProof::from_bytes: before verify() is entered */
#define BPP_TIER_DEGREE 2           /* extension degree of every item (:637-659) */
#define BPP_TIER_PROMISE 3          /* minimum-value promises (:674-682) */
#define BPP_TIER_STATEMENT_POINT 4  /* a commitment does not decode (a RangeStatement holds points) */
#define BPP_TIER_PASS1 5            /* transcript replay of ANY proof (:816-850) */
#define BPP_TIER_PASS2 6            /* per proof, in proof order: decompression, L/R count (:859-888) */
#define BPP_TIER_MSM 7              /* the final multiscalar check (:1057-1062) */
#define BPP_TIER_ENGINE 255         /* an engine fault (negative code) on some rank */

/* VerifyAction (src/range_proof.rs:46-54) */
#define BPP_VERIFY_ONLY 0
#define BPP_RECOVER_AND_VERIFY 1
#define BPP_RECOVER_ONLY 2

/* MAX_RANGE_PROOF_BATCH_SIZE (src/range_proof.rs:76): pass as `chunk` for reference-sized batches */
#define BPP_REFERENCE_CHUNK 256

typedef struct bpp_ctx bpp_ctx;

/* ---- context ---- */
int bpp_ctx_create(bpp_ctx **out, int device_id);
/* same, but all work is enqueued on the caller's hipStream_t (e.g. torch's current stream) */
int bpp_ctx_create_on_stream(bpp_ctx **out, int device_id, void *hip_stream);
void bpp_ctx_destroy(bpp_ctx *ctx);
const char *bpp_ctx_last_error(bpp_ctx *ctx);
/* per-context knobs for tests and A/B timing; value -1 restores the engine's own rule.  Names: "transcripts_wave",
 * "tables_wave", "side_decompress", "msm_quad", "msm_final_quad" (0 / 1: force the one-lane or the latency form of that stage),
 * "msm_c_bias" (extra MSM window bits of small calls), "msm_c_max" (widest MSM window), "msm_c_add", "msm_rc2" (0: one window per wavefront in the bucket reduction), "msm_split",
 * "fb_threads", "prove_subs", "prove_fused" (0: three launches per prover round instead of one), "prove_prio" (1: the prover's small
 * kernels on a high-priority stream), "fused_columns" (0: per-proof generator rows + k_reduce_static instead of the column sums inside
 * k_scalars_lanes), "static_gemm" (1 / 0: those column sums as ONE integer matrix product over the proofs of a group on the
 * matrix cores, kernels_static_gemm.h, or by Montgomery products per (proof, generator); by itself the engine takes the matrix
 * product from aggregation 8 on), "lazy_columns" (0: every product of the column sums inside k_scalars_lanes reduced by itself
 * instead of one reduction per workgroup and column), "ct" (which of the SECRET-ONLY terms -- those the reference computes in
 * constant time -- take the uniform-access forms of csrc/ct.h, where no address and no branch depends on the scalar: 1 = the default:
 * bpp_pedersen_commit and the prover's witness check (src/generators/pedersen_gens.rs:112-122, src/range_proof.rs:275-284);
 * 2: A1 and B of the final round as well (:572-584; a 256-doubling ladder on the call's last stretch, see DESIGN.md for its
 * measured cost); 0: everything through the fixed-base tables, addressed by the scalars' digits).  The environment variables BPP_<NAME> give
 * "chain" (where the batch-weight chains of a verification run -- src/range_proof.rs:811,849,853,894, one strictly sequential
 * sponge per reference batch: 0 = on host cores (csrc/chain_host.h: lowest latency, 0.04 host-core-ms per 1024 proofs), 1 = on
 * the device, one wavefront per reference batch behind PASS 1 (csrc/chain_dev.h: the calling thread only enqueues and the
 * rank needs no host cores in proportion to its throughput; a zero weight, probability 2^-252, sends the call through the host
 * chains once more, which redraw as the reference does), 2 = the sponges on host cores, Scalar::from_bytes_mod_order_wide and the
 * look for a zero weight on the device (half the host time of 0), -1 = the engine's rule: 2 for calls of 4096 proofs and more, else
 * 0), "wait" (how the calling thread waits for the device: 0 = hipStreamSynchronize, which spins on a core; 1 = naps of 50 us between
 * looks at an event, 1-2 % of a core; -1 = the engine's rule: naps for calls of 4096 proofs and more).  The environment variables BPP_<NAME> give
 * the initial values and are read ONCE, when the context is created: no verification path calls getenv. */
int bpp_ctx_set_option(bpp_ctx *ctx, const char *name, int value);
/* verifications of this context whose weights were made on the device ("chain" 1 or 2), and how many of those ran once more
 * with everything on the host because a weight came out zero */
int bpp_device_chain_stats(bpp_ctx *ctx, uint64_t *calls, uint64_t *redraws);

/* ---- runtime preconditions and the admission gate for small calls ("Runtime preconditions") ----
 * Separate callers of RangeProof::verify_batch hand over at most MAX_RANGE_PROOF_BATCH_SIZE = 256 proofs per call
 * (src/range_proof.rs:73-76,712-752).  Such a call is a chain of latency-bound kernels; the chip runs about six of them side
 * by side, and with many more in flight (every context brings a stream and a side stream onto the runtime's hardware queues)
 * all of them get slower.  The library therefore admits at most `small_call_limit` calls of <= 1024 proofs per device at a
 * time (default 12, environment BPP_SMALL_CALLS_IN_FLIGHT, 0 = no gate); the others wait their turn on the host in arrival
 * order.  Large calls pass freely.  bpp_small_call_limit sets the limit (limit < 0: query only) and returns the previous one.
 * bpp_runtime_info_get reports what the library sees: live contexts of the device, the hardware queues the HIP runtime uses
 * (GPU_MAX_HW_QUEUES as read at ITS start-up; 4 when unset -- a library cannot change it for its host), the host pool, the
 * gate's counters.  When a new context makes contexts > hw_queues, bpp_ctx_create still succeeds and leaves a note where
 * bpp_ctx_last_error finds it. */
typedef struct {
  int device;
  uint32_t contexts, contexts_peak; /* live / most ever: each owns a stream, small inputs use a second one */
  uint32_t hw_queues;               /* GPU_MAX_HW_QUEUES (4 when unset) */
  uint32_t host_threads;            /* bpp_host_threads() */
  uint32_t small_call_limit, small_calls_in_flight;
  uint64_t small_calls, small_calls_queued; /* gated calls so far; those that had to wait */
  uint32_t oversubscribed;          /* 1: contexts > hw_queues */
} bpp_runtime_info;
int bpp_runtime_info_get(bpp_ctx *ctx, bpp_runtime_info *out);
int bpp_small_call_limit(bpp_ctx *ctx, int limit);

/* ---- B1: multiscalar traits ----
 * bpp_precomp_create  = VartimePrecomputedMultiscalarMul::new(static_points)   (src/generators/bulletproof_gens.rs:103)
 * bpp_msm_mixed       = ::vartime_mixed_multiscalar_mul(static_scalars, dyn_scalars, dyn_points)
 *                        (src/range_proof.rs:339-345, :1050-1057); n_static <= table size, rest = zero padding
 * bpp_msm_vartime     = VartimeMultiscalarMul::vartime_multiscalar_mul          (src/range_proof.rs:482-495, :512-521)
 *                        also serves MultiscalarMul::multiscalar_mul            (src/generators/pedersen_gens.rs:120)
 * A point that does not decode makes the call return BPP_ERR_INVALID_ARGUMENT. */
int bpp_precomp_create(bpp_ctx *ctx, const uint8_t *points32, size_t count, uint64_t *handle);
int bpp_precomp_destroy(bpp_ctx *ctx, uint64_t handle);
/* Arc::clone: `ctx` takes its own reference on a handle created by another context of the same device (dropped by
 * bpp_precomp_destroy(ctx, handle) or when ctx is destroyed) */
int bpp_precomp_retain(bpp_ctx *ctx, uint64_t handle);
int bpp_msm_mixed(bpp_ctx *ctx, uint64_t handle, const uint8_t *static_scalars32, size_t n_static,
                  const uint8_t *dyn_scalars32, const uint8_t *dyn_points32, size_t n_dyn, uint8_t out_point32[32]);
int bpp_msm_vartime(bpp_ctx *ctx, const uint8_t *scalars32, const uint8_t *points32, size_t n,
                    uint8_t out_point32[32]);
/* many independent MSMs in one launch: group g covers terms [group_off[g], group_off[g+1]) */
int bpp_msm_vartime_batched(bpp_ctx *ctx, const uint8_t *scalars32, const uint8_t *points32,
                            const uint32_t *group_off, size_t n_groups, uint8_t *out_points32 /* n_groups x 32 */);

/* ---- B2: parameters = RangeParameters::init + BulletproofGens::new + PedersenGens ----
 * (src/range_parameters.rs:32-58, src/generators/bulletproof_gens.rs:83-112, src/ristretto.rs:67-112)
 * h_base32 / g_bases32 may be NULL: the reference's defaults (Ristretto basepoint; SHA3-512 hash-to-group of
 * "RISTRETTO_MASKING_BASEPOINT_k") are derived on the device. */
int bpp_params_create(bpp_ctx *ctx, uint32_t bit_length, uint32_t max_aggregation, uint32_t extension_degree,
                      const uint8_t *h_base32, const uint8_t *g_bases32, uint64_t *params);
int bpp_params_destroy(bpp_ctx *ctx, uint64_t params);
/* Arc::clone of a RangeParameters object (src/range_parameters.rs:20-30 holds Arc'd generators): `ctx` takes its own
 * reference on `params`, whichever context created it; one set of device tables (generators, and the prover's
 * fixed-base windows once built) serves every holder */
int bpp_params_retain(bpp_ctx *ctx, uint64_t params);
/* compressed generators, party-major like gi_base_iter()/hi_base_iter() (src/range_parameters.rs:99-106):
 * gi_out32, hi_out32: bit_length*max_aggregation x 32; h_out32: 32; g_out32: extension_degree x 32. Any may be NULL. */
int bpp_params_export(bpp_ctx *ctx, uint64_t params, uint8_t *gi_out32, uint8_t *hi_out32, uint8_t *h_out32,
                      uint8_t *g_out32);
/* PedersenGens::commit for `count` openings (src/generators/pedersen_gens.rs:112-122):
 * values[count], blindings32[count][n_blind][32] -> commitments32[count][32]; 1 <= n_blind <= extension degree */
int bpp_pedersen_commit(bpp_ctx *ctx, uint64_t params, const uint64_t *values, const uint8_t *blindings32,
                        uint32_t n_blind, size_t count, uint8_t *commitments32);

/* ---- B2: batch verification = RangeProof::verify_batch / verify ---- */
typedef struct {
  const uint8_t *proof;         /* RangeProof::to_bytes() (src/range_proof.rs:1120-1150) */
  size_t proof_len;
  const uint8_t *commitments32; /* statement.commitments_compressed, m x 32 (src/range_statement.rs:27) */
  uint32_t m;                   /* aggregation factor of this statement */
  const uint64_t *min_values;   /* m entries; value ignored where min_present[j] == 0 */
  const uint8_t *min_present;   /* m entries, Option<u64>::is_some; NULL = all None */
  const uint8_t *seed_nonce32;  /* NULL = None (src/range_statement.rs:31) */
  /* the caller's merlin::Transcript: EITHER its 203-byte STROBE state (200 state bytes, pos, pos_begin, cur_flags)
   * OR, for a fresh Transcript::new(label), the label */
  const uint8_t *transcript_state;
  const uint8_t *transcript_label;
  size_t label_len;
} bpp_verify_item;

/* One call = RangeProof::verify over every `chunk` consecutive proofs (chunk == 0: the whole batch is ONE
 * reference batch; chunk == 256 reproduces verify_batch's chunking but -- unlike src/range_proof.rs:740-751 --
 * verifies EVERY chunk, first failing chunk's error wins).
 * masks_out: n_items x extension_degree x 32 (may be NULL for BPP_VERIFY_ONLY); mask_present: n_items (may be NULL).
 * errbuf receives the reference's informational message text. */
int bpp_verify_batch(bpp_ctx *ctx, uint64_t params, const bpp_verify_item *items, size_t n_items, int action,
                     size_t chunk, uint8_t *masks_out, uint8_t *mask_present, char *errbuf, size_t errbuf_len);

/* Variant for callers that keep merlin on their side (SURVEY 8b option (i)): the Fiat-Shamir replay of PASS 1
 * (src/range_proof.rs:816-850) is done by the caller, who passes per proof
 *   challenges32[i] : (rounds_i + 3) x 32 canonical scalars  y, z, e_0 .. e_{rounds-1}, e_final   (:833-842)
 *   rng_out32       : n_items x 32, the 32 bytes drawn from transcript.to_verifier_rng(...)         (:845-848)
 * and the engine runs everything else (weight chain, decompression, PASS 2, MSM, mask recovery).  The transcript
 * fields of the items are ignored.  Identity-encoded proof members and zero challenges are still rejected. */
int bpp_verify_batch_with_challenges(bpp_ctx *ctx, uint64_t params, const bpp_verify_item *items, size_t n_items,
                                     const uint8_t *const *challenges32, const uint8_t *rng_out32, int action, size_t chunk,
                                     uint8_t *masks_out, uint8_t *mask_present, char *errbuf, size_t errbuf_len);

/* Same in two steps so a batch can stay resident in HBM: upload parses/validates/packs and copies to the device;
 * verify_resident runs only device work plus the (inherently sequential) weight chain. */
int bpp_batch_upload(bpp_ctx *ctx, uint64_t params, const bpp_verify_item *items, size_t n_items, uint64_t *batch,
                     char *errbuf, size_t errbuf_len);
int bpp_batch_destroy(bpp_ctx *ctx, uint64_t batch);
int bpp_verify_resident(bpp_ctx *ctx, uint64_t batch, int action, size_t chunk, uint8_t *masks_out,
                        uint8_t *mask_present, char *errbuf, size_t errbuf_len);
/* optional: do the one-off work of the first bpp_verify_resident(batch, ..., chunk) now -- group layout, MSM plan and
 * the allocation of every work buffer of that plan (~15 hipMallocs) -- so that no verification call pays for it */
int bpp_batch_prepare(bpp_ctx *ctx, uint64_t batch, size_t chunk);

/* ---- packed form of a homogeneous batch: what a caller of RangeProof::verify_batch(&[Transcript], &[RangeStatement],
 * &[RangeProof]) (src/range_proof.rs:712-717) holding statements and proofs of ONE shape serialises them into -- no
 * per-item pointer structs.  All proofs have the same length, all statements the same aggregation factor, one transcript
 * serves every item (benches/range_proof.rs:98, tests/ristretto.rs:225).  Same checks, same error kinds and precedence as
 * the item form (one routine checks both); results are bit-identical (tests/test_gpu_packed.py). */
typedef struct {
  size_t n_items;
  const uint8_t *proofs;           /* proof i = RangeProof::to_bytes() at proofs + i * proof_stride */
  size_t proof_len;                /* length of every proof */
  size_t proof_stride;             /* >= proof_len; == proof_len when the proofs lie back to back */
  const uint8_t *commitments32;    /* n_items x m x 32 */
  uint32_t m;                      /* aggregation factor of every statement */
  const uint64_t *min_values;      /* n_items x m */
  const uint8_t *min_present;      /* n_items x m, Option<u64>::is_some; NULL = all None */
  const uint8_t *seed_nonces32;    /* n_items x 32; NULL = all None */
  const uint8_t *seed_present;     /* n_items; NULL = every item has a nonce (when seed_nonces32 != NULL) */
  const uint8_t *transcript_state; /* 203-byte STROBE state, or NULL: Transcript::new(transcript_label) */
  const uint8_t *transcript_label;
  size_t label_len;
} bpp_packed_batch;
int bpp_batch_upload_packed(bpp_ctx *ctx, uint64_t params, const bpp_packed_batch *in, uint64_t *batch, char *errbuf,
                            size_t errbuf_len);
int bpp_verify_batch_packed(bpp_ctx *ctx, uint64_t params, const bpp_packed_batch *in, int action, size_t chunk,
                            uint8_t *masks_out, uint8_t *mask_present, char *errbuf, size_t errbuf_len);

/* Pipelined host-buffers-in verification inside ONE context.  submit parses, validates and packs `in` into page-locked
 * staging on the calling thread (construction errors are returned by submit itself, as RangeStatement::init /
 * RangeProof::from_bytes would raise them before verify_batch is entered) and returns; the caller's buffers are free again.
 * DMA, kernels and the weight chain of that call then run on one of the context's `depth` internal lanes (own stream, own
 * staging, own work buffers; default depth 3, bpp_ctx_pipeline_depth before the first submit changes it) while the caller
 * submits the next batch: upload k+1 overlaps verify k.  collect blocks until the ticket's call is done and returns exactly
 * what bpp_verify_batch_packed would have returned.  Tickets may be collected in any order; every ticket must be collected
 * (bpp_ctx_destroy waits for the ones in flight).  submit blocks while all lanes are busy. */
int bpp_ctx_pipeline_depth(bpp_ctx *ctx, uint32_t depth);
int bpp_verify_submit_packed(bpp_ctx *ctx, uint64_t params, const bpp_packed_batch *in, int action, size_t chunk,
                             uint64_t *ticket, char *errbuf, size_t errbuf_len);
int bpp_verify_collect(bpp_ctx *ctx, uint64_t ticket, uint8_t *masks_out, uint8_t *mask_present, char *errbuf,
                       size_t errbuf_len);

/* ---- phased form of the same verification, for sharding one reference batch across GPUs ----
 * phase1: PASS 1 of verify (src/range_proof.rs:816-850) for this rank's proofs -> the 32 transcript-RNG bytes per
 *         proof that feed the weight transcript (:845-849).
 * weights_from_chain: the weight transcript itself (:811,:849,:853,:894) over the WHOLE batch in global proof order
 *         (pure host function: the chain is a sequential sponge).
 * phase2: PASS 2 (:856-1033) + this rank's share of the final MSM (:1050) -> partial accumulator as 128 bytes
 *         (X,Y,Z,T canonical field elements).  Only rank 0 (include_pedersen != 0 on exactly one rank is NOT needed:
 *         every rank adds its own share of the g/h base scalars).
 * accumulators_sum_is_identity: sum of all ranks' accumulators == identity  (:1057). */
int bpp_verify_phase1(bpp_ctx *ctx, uint64_t batch, uint8_t *rng_out32 /* n_items x 32 */, char *errbuf,
                      size_t errbuf_len);
int bpp_weights_from_chain(const uint8_t *rng32_all, size_t n_total, uint8_t *weights32_out /* n_total x 32 */);
/* n_groups independent weight transcripts of n_per_group proofs each (what verify_resident runs for chunk > 0): same
 * result as n_groups calls of bpp_weights_from_chain, but the chains advance in lockstep on vector Keccak (AVX-512: 8,
 * AVX2: 4 chains per core) and bundles are spread over host threads. */
int bpp_weights_from_chains(const uint8_t *rng32_all, size_t n_groups, size_t n_per_group, uint8_t *weights32_out);
int bpp_verify_phase2(bpp_ctx *ctx, uint64_t batch, const uint8_t *weights32 /* n_items x 32 */,
                      uint8_t accumulator128[128], char *errbuf, size_t errbuf_len);
int bpp_accumulators_sum_is_identity(bpp_ctx *ctx, const uint8_t *accumulators128, size_t n, int *is_identity);

/* ---- ONE reference batch sharded over the GPUs of a node, behind the C ABI (BASELINE configs[3]) ----
 * One process (or thread) per GPU; rank r holds `counts[r]` consecutive proofs of the batch as a resident batch on its own
 * context.  bpp_verify_sharded runs RangeProof::verify (src/range_proof.rs:756-1065) over the union:
 *   PASS 1 on every rank's own proofs, decompression and the weight-free scalars right behind it
 *   -> RCCL all_gather of the 32 transcript-RNG bytes per proof (device buffers, no host hop) as soon as PASS 1 is done
 *   -> the batch-weight transcript (:811,:849,:853,:894) replayed by every rank over ALL proofs in order (a sequential
 *      sponge: cheaper to replay than to broadcast), each rank keeps the weights of its own proofs
 *   -> PASS 2 + the rank's share of the final MSM (:1050) -> one accumulator point per rank
 *   -> RCCL all_gather of the 128-byte accumulators, each with the 128-byte finding of its rank (RCCL has no group-law
 *      reduction, so the north star's "all-reduce of the accumulator" is gather + the same sum on every rank), sum and
 *      identity test (:1057) on the device; a finding of any rank comes before the final check, as in verify().
 * EVERY rank reaches both collectives whatever it found locally, and every rank returns the same result: the error the
 * single-process verify() would have raised first, decided by numeric tier (BPP_TIER_*), then lowest rank.
 * Return value: 0 Ok, 1..5 ProofError kind (same on every rank), BPP_ERR_COMM when RCCL fails (library missing,
 * communicator error: the communicator must then be destroyed), other negatives = engine fault on some rank.
 * VerifyOnly semantics (masks are per proof and need no exchange: recover them with bpp_verify_resident on the shard).
 *
 * Communicators: bpp_comm_unique_id (rank 0) -> the caller ships the 128 bytes to the other ranks over its own channel ->
 * bpp_comm_create on every rank (ncclCommInitRank: collective); or bpp_comm_adopt of an ncclComm_t the caller already
 * has; or bpp_comm_create_callbacks with the caller's own all_gather.  A communicator serialises its own calls; use one per
 * thread that verifies concurrently.
 *
 * bpp_comm_adopt does NOT take ownership: the ncclComm_t stays the caller's (it may be a framework's process-group communicator
 * in use elsewhere), bpp_comm_destroy never destroys it, and a collective that misses its deadline (bpp_comm_set_timeout) never
 * aborts it -- the handle is marked dead and the call returns BPP_ERR_COMM, and it is the OWNER who calls ncclCommAbort (which
 * also lets the collective still spinning on this rank exit).  Only communicators made by bpp_comm_create are aborted and
 * destroyed by this library. */
typedef struct bpp_comm bpp_comm;
int bpp_comm_unique_id(uint8_t id128[128]);
int bpp_comm_create(bpp_ctx *ctx, const uint8_t id128[128], int rank, int world, bpp_comm **out);
int bpp_comm_adopt(bpp_ctx *ctx, void *nccl_comm, int rank, int world, bpp_comm **out);
/* The caller's own transport instead of RCCL (its MPI / TCP / gloo channel; also how several ranks share ONE GPU, which RCCL
 * refuses): `all_gather` receives this rank's `bytes_per_rank` bytes in HOST memory and must fill recv[r * bytes_per_rank ...]
 * with rank r's block for every r (its own included), in the SAME order of calls on every rank; it blocks until that is done and
 * returns 0, or non-zero on failure (the call then returns BPP_ERR_COMM and the handle is dead).  It is called on the thread
 * that made the bpp_verify_sharded* call, two or three times per call (32 bytes per proof; weights when the chains are shared
 * out; 256 bytes per batch).  Any deadline is the transport's own: bpp_comm_set_timeout cannot interrupt a callback. */
typedef int (*bpp_all_gather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
int bpp_comm_create_callbacks(bpp_ctx *ctx, int rank, int world, bpp_all_gather_fn all_gather, void *user, bpp_comm **out);
/* In-process stand-in for tests on a single GPU: the `world` ranks are threads of one process, each with its own context on
 * the same device; an all_gather is a rendezvous of the threads plus device-to-device copies.  Everything else of
 * bpp_verify_sharded(_wave) is the code the RCCL form runs.  Ranks of one group pass the same (arbitrary) group_id. */
int bpp_comm_create_local(bpp_ctx *ctx, uint64_t group_id, int rank, int world, bpp_comm **out);
void bpp_comm_destroy(bpp_comm *comm);
const char *bpp_comm_last_error(bpp_comm *comm);
/* Deadline of every wait for a collective on this communicator, in ms (default 60 000, or BPP_COMM_TIMEOUT_MS at creation;
 * 0 = wait for ever; a BPP_COMM_TIMEOUT_MS that is not a number keeps the default and leaves a note in bpp_ctx_last_error).  A
 * peer that died or never made the call would leave the all_gather spinning on this rank for ever: when the deadline passes
 * the call returns BPP_ERR_COMM on every surviving rank, and so does every later call on the handle -- destroy it and build a
 * new one over the ranks that are left.  A communicator made by bpp_comm_create is aborted (ncclCommAbort) at that point; an
 * ADOPTED one is left alone, see bpp_comm_adopt. */
int bpp_comm_set_timeout(bpp_comm *comm, uint32_t timeout_ms);
int bpp_verify_sharded(bpp_comm *comm, bpp_ctx *ctx, uint64_t batch, const uint32_t *counts /* world entries */,
                       int *tier_out, int *rank_out, char *errbuf, size_t errbuf_len);
/* A WAVE of k independent sharded batches (batch i resident on ctxs[i], every rank passes its shards of the same k batches
 * in the same order, all with the same `counts`): their PASS-1 work runs concurrently on the k contexts' streams, ONE
 * coalesced all_gather carries the RNG bytes of all k, the k weight chains run side by side on the host pool, ONE all_gather
 * carries the k accumulators.  results[i] receives batch i's outcome (code as bpp_verify_sharded returns it, tier, rank,
 * message).  Return value: 0 = the wave ran (look at results), BPP_ERR_COMM / negative = the wave itself failed. */
typedef struct {
  int code, tier, rank;
  uint32_t index;       /* position in the whole batch of the proof the finding belongs to (tiers checked per proof) */
  char msg[160];
} bpp_shard_result;
int bpp_verify_sharded_wave(bpp_comm *comm, bpp_ctx *const *ctxs, const uint64_t *batches, size_t k, const uint32_t *counts,
                            bpp_shard_result *results);
/* The GROUPED form: this rank's shards of n_groups independent reference batches (same `counts` for all) are ONE resident
 * batch on ONE context -- group g = proofs [g c, (g+1) c) of it, c = counts[rank] > 0 -- so every verifier kernel is
 * launched once for all groups (as bpp_verify_resident does with chunk = c) and each of the two all_gathers carries all
 * groups.  results[g] = group g's outcome, exactly as a bpp_verify_sharded call on that batch alone would give it.  The form
 * for many batches with small shards: a wave of k contexts pays a dozen launches and a stream per batch, this pays them
 * once.  With several ranks the weight chains are shared out (rank r replays those of groups r, r + world, ...) and a third
 * all_gather hands every rank all weights: the sequential replay, not the GPUs, bounds the rate when every rank replays
 * every chain.  src/range_proof.rs:712-752 (one reference batch per group), :811-853 (its weight chain over all ranks' proofs). */
int bpp_verify_sharded_groups(bpp_comm *comm, bpp_ctx *ctx, uint64_t batch, size_t n_groups, const uint32_t *counts,
                              bpp_shard_result *results /* n_groups */);
/* k grouped batches (batch i resident on ctxs[i], every one holding n_groups x counts[rank] proofs) as a software pipeline of
 * ONE host thread on ONE communicator: phase 1 of all k is enqueued; then, batch by batch, the first exchange, the weight
 * chains and the enqueueing of phase 2 (while the other batches' kernels keep the GPU busy); then, batch by batch, the findings
 * and the second exchange.  Every rank issues its collectives in the same order by construction, which calls from several
 * host threads on several communicators cannot promise.  results: k x n_groups, batch-major. */
int bpp_verify_sharded_groups_wave(bpp_comm *comm, bpp_ctx *const *ctxs, const uint64_t *batches, size_t k, size_t n_groups,
                                   const uint32_t *counts, bpp_shard_result *results /* k x n_groups */);

/* ---- pooled small calls ----
 * Reference batches of different sizes as the groups of ONE engine call: group g = proofs [group_first[g], group_first[g+1])
 * of the resident batch (group_first[0] = 0, group_first[n_groups] = the batch size), every group verified as its own
 * verify() (VerifyOnly) with its own outcome in results[g] (code, tier, index inside the group, message; rank = -1). */
int bpp_verify_resident_groups(bpp_ctx *ctx, uint64_t batch, const uint32_t *group_first, size_t n_groups, bpp_shard_result *results);
/* The same with a VerifyAction PER GROUP (src/range_proof.rs:46-54; actions == NULL: VerifyOnly everywhere) and the masks of
 * the groups that recover them (:941-969): masks_out n_items x extension_degree x 32, mask_present n_items (either may be
 * NULL).  A group gets its masks only where its own verify() would have returned Ok; every other slot is zero / absent.  A
 * RecoverOnly group is never held to the final check (:1040-1043); when every group is RecoverOnly neither the weight chains
 * nor PASS 2 run. */
int bpp_verify_resident_groups_actions(bpp_ctx *ctx, uint64_t batch, const uint32_t *group_first, size_t n_groups, const int *actions,
                                       bpp_shard_result *results, uint8_t *masks_out, uint8_t *mask_present);
/* bpp_batcher: many host threads, each with ONE reference batch per call.  Separate small calls stop at about 5 000 calls per
 * second whatever the number of callers (a small call is a chain of latency-bound kernels and the chip runs about six of those
 * side by side); the batcher pools the calls that are waiting into grouped engine calls (bpp_verify_resident_groups) on
 * `lanes` contexts of its own (the first one is `ctx`; 0 = the default of two), without a thread of its own: whichever caller finds a lane free
 * leads the next pooled call for everybody queued behind it.  bpp_batcher_verify blocks and returns exactly what
 * bpp_verify_batch_packed(ctx, params, in, BPP_VERIFY_ONLY, 0, ...) would: 0 or the ProofError kind, message in errbuf.
 * `shape` fixes what can be pooled (proof_len, m, transcript label; other inputs are verified on their own).  max_wait_us:
 * how long a leader waits for company (0: takes what is there); max_calls: most batches per pooled call (0: 64). */
typedef struct bpp_batcher bpp_batcher;
int bpp_batcher_create(bpp_ctx *ctx, uint64_t params, const bpp_packed_batch *shape, uint32_t lanes, uint32_t max_wait_us, uint32_t max_calls,
                       bpp_batcher **out);
int bpp_batcher_verify(bpp_batcher *b, const bpp_packed_batch *in, char *errbuf, size_t errbuf_len);
/* Any VerifyAction through the pool (round 4): returns exactly what bpp_verify_batch_packed(ctx, params, in, action, 0,
 * masks_out, mask_present, ...) would -- the caller's own seed nonces in (in->seed_nonces32), its own masks out.  Calls pool
 * whatever their shape (proof length, aggregation factor, transcript label or state: the pooled upload goes through the item
 * form, out of the callers' buffers); RecoverOnly calls (a wallet scanning outputs: no final check, :1040-1043) pool among
 * themselves.  `shape` of bpp_batcher_create is no longer needed and may be NULL.  Secrets: nonces are read where the caller
 * keeps them, masks pass through the lane's buffers, which are wiped before the lane is handed on. */
int bpp_batcher_verify_action(bpp_batcher *b, const bpp_packed_batch *in, int action, uint8_t *masks_out, uint8_t *mask_present,
                              char *errbuf, size_t errbuf_len);
/* most requests (0: keep) and most proofs (0: keep; default 16384) of one pooled call */
int bpp_batcher_set_limits(bpp_batcher *b, uint32_t max_calls, uint32_t max_proofs);
/* the largest pooled call so far: requests and proofs in it (never above the limits) */
int bpp_batcher_largest_pool(bpp_batcher *b, uint32_t *calls, uint32_t *proofs);
int bpp_batcher_stats(bpp_batcher *b, uint64_t *pooled_calls, uint64_t *engine_calls, uint64_t *solo_calls);
void bpp_batcher_destroy(bpp_batcher *b); /* waits for the calls in flight; no call may start once it has been called */
/* host wall-clock split of the last wave on `comm` (ms): enqueueing phase 1 on the k streams, the first exchange (waits for
 * PASS 1 only: all_gather, RNG bytes down), the k weight chains, enqueueing phase 2, waiting for the k streams, the second
 * exchange with the sum and identity test (wait1_ms is always 0 since the first exchange no longer waits for all of phase 1) */
typedef struct {
  float enqueue1_ms, wait1_ms, gather1_ms, chains_ms, enqueue2_ms, wait2_ms, gather2_ms;
  uint32_t batches;
} bpp_shard_timing;
int bpp_comm_last_timing(bpp_comm *comm, bpp_shard_timing *out);
/* host-only pieces of the above, exported for callers that bring their own transport and for the CPU tests
 * (tests/test_dist_gloo.py): the 128-byte finding a rank contributes, and the rule every rank applies to the gathered ones.
 *   bpp_shard_local_trailer: first finding of a rank's n proofs from the per-proof facts -- defer[i] (bit 0: extension
 *     degree differs, bit 1: promise too large; may be NULL), status[i] (bit 0 PASS-1 failure, bit 1 proof point does not
 *     decode, bit 2 commitment does not decode), rounds_bad[i] (0, 3 = InvalidLength, 5 = SizeOverflow) -- in the
 *     reference's order of checks; first_index = position of the rank's first proof in the whole batch
 *   bpp_shard_trailer: a finding given directly (tier, code, index, message)
 *   bpp_shard_resolve: trailer of rank r at trailers + r * stride -> the winning finding; returns its code (0: all clean) */
#define BPP_SHARD_TRAILER_BYTES 128
int bpp_shard_local_trailer(const uint8_t *defer, const uint32_t *status, const uint8_t *rounds_bad, uint32_t n,
                            uint32_t first_index, uint8_t trailer_out[BPP_SHARD_TRAILER_BYTES]);
int bpp_shard_trailer(int tier, int code, uint32_t index, const char *msg, uint8_t trailer_out[BPP_SHARD_TRAILER_BYTES]);
int bpp_shard_resolve(const uint8_t *trailers, size_t stride, int world, int *tier_out, int *rank_out, uint32_t *index_out,
                      char *errbuf, size_t errbuf_len);

/* ---- B2: batch prover = RangeProof::prove_with_rng (src/range_proof.rs:232-608), one call for n independent proofs ----
 * Every item is one (statement, witness, transcript, rng) quadruple of the reference API:
 *   values/blindings32  RangeWitness: per opening v (u64) and r[0..t) (src/commitment_opening.rs:14-37)
 *   commitments32       statement.commitments_compressed; each must equal commit(v_j, r_j) (:275-284)
 *   min_values/min_present, seed_nonce32   as in bpp_verify_item (src/range_statement.rs:21-32)
 *   transcript_*        the caller's merlin::Transcript (state or Transcript::new(label))
 *   rng_bytes           what the external `rng` would have returned: (rounds + 3) draws of 32 bytes, rounds = log2(m * n)
 *                       (the reference pulls exactly that many through TranscriptRngBuilder::finalize)
 * All items of one call must share the aggregation factor m.  Output: to_bytes() of each proof, 1 + 32*(t + 5 + 2*rounds)
 * bytes, written at proofs_out + i * proof_stride; *proof_len receives that length.  Same inputs -> same bytes as the
 * reference (any equivalent schedule yields identical canonical encodings). */
typedef struct {
  const uint64_t *values;        /* m */
  const uint8_t *blindings32;    /* m x t x 32 */
  const uint8_t *commitments32;  /* m x 32 */
  uint32_t m;
  const uint64_t *min_values;
  const uint8_t *min_present;    /* NULL = all None */
  const uint8_t *seed_nonce32;   /* NULL = None; only with m == 1 */
  const uint8_t *transcript_state;
  const uint8_t *transcript_label;
  size_t label_len;
  const uint8_t *rng_bytes;
  size_t rng_len;                /* >= (rounds + 3) * 32 */
} bpp_prove_item;
int bpp_prove_batch(bpp_ctx *ctx, uint64_t params, const bpp_prove_item *items, size_t n_items, uint8_t *proofs_out,
                    size_t proof_stride, size_t *proof_len, char *errbuf, size_t errbuf_len);

/* ---- parity / diagnostics: intermediates of the last verify on `batch`, for differential tests ---- */
#define BPP_TRACE_CHALLENGES 1     /* per proof (rmax+3) x 32: y, z, e_0.., e_final (canonical), rmax = bpp_batch_shape's max_rounds
                                      (largest round count in the batch, at most 11: proofs claiming more are refused) */
#define BPP_TRACE_RNG_OUT 2        /* n x 32 */
#define BPP_TRACE_WEIGHTS 3        /* n x 32 */
#define BPP_TRACE_STATIC_SCALARS 4 /* groups x (2*max_mn + t + 1) x 32: gi0,hi0,gi1,hi1,...,g_0..g_{t-1},h */
#define BPP_TRACE_DYNAMIC_SCALARS 5 /* total_dyn x 32, proof order: C_j.., A1, B, A, L.., R.. */
#define BPP_TRACE_MSM_RESULT 6     /* groups x 32 compressed */
#define BPP_TRACE_PLAN 7           /* four uint32: bit 0 generator columns summed in k_scalars_lanes, bit 1 taken from k_static_gemm;
                                      proofs per workgroup of k_scalars_lanes; K chunks of k_static_gemm; groups */
int bpp_batch_trace(bpp_ctx *ctx, uint64_t batch, int what, uint8_t *out, size_t out_len, size_t *written);
/* shape helpers for the above */
int bpp_batch_shape(bpp_ctx *ctx, uint64_t batch, uint32_t *n_items, uint32_t *max_rounds, uint32_t *max_mn,
                    uint32_t *total_dyn, uint32_t *groups);

/* ---- per-stage device timing of the last verify (hipEvents on the ctx stream) ---- */
typedef struct {
  float transcripts_ms, decompress_ms, chain_host_ms, scalars_ms, reduce_ms;
  float msm_digits_ms, msm_sort_ms, msm_accumulate_ms, msm_bucket_reduce_ms, msm_final_ms;
  float total_ms;
  uint32_t msm_terms, msm_window_bits, msm_windows, msm_groups;
  float masks_ms; /* k_masks (mask recovery, src/range_proof.rs:941-969); 0 for VerifyOnly */
  float chain_device_ms; /* the weight chains as kernels (option "chain" = 1: k_weight_chain + k_chain_finish); 0 with the chains on the host */
} bpp_profile;
/* on = 1: an event at every stage boundary (thirteen per verification);
 * on = 2: the two events around the roofline kernel only (msm_accumulate_ms; every other interval reads 0) */
int bpp_profile_enable(bpp_ctx *ctx, int on);
int bpp_profile_get(bpp_ctx *ctx, bpp_profile *out);

/* the batch prover's dominant kernel (k_fb_msm, fixed-base MSM) over the last bpp_prove_batch of this ctx: summed event
 * time of its launches, the terms it consumed (witness check, L/R of every round, A1/B), table geometry */
typedef struct {
  float fb_msm_ms, total_ms;
  uint64_t fb_terms;
  uint32_t fb_launches, fb_window_bits, fb_windows, sub_batches;
} bpp_prove_profile;
int bpp_prove_profile_get(bpp_ctx *ctx, bpp_prove_profile *out);

/* size of the process-wide host worker pool that runs the batch-weight chains and the upload packer
 * (BPP_HOST_THREADS, default min(usable cores, 32), usable = affinity mask capped by the cgroup CPU quota): the verifier's throughput depends on it */
int bpp_host_threads(void);
/* CPU time (ns, thread clocks, summed over the pool and the calling threads) this process has spent in the library's host jobs
 * so far: the batch weight chains (src/range_proof.rs:849-853,894 -- the one sequential part of a verification, kept on the host)
 * and the parsing of uploaded batches.  The difference over a run / (steps x step time) = host cores a rank keeps busy: with
 * several ranks on one node it tells a host-bound scaling curve from a GPU-bound one. */
uint64_t bpp_host_pool_cpu_ns(void);

/* diagnostics: the shader clock the device holds RIGHT NOW, sampled by one napping wavefront on this context's stream for
 * about `window_us` microseconds (s_memtime against the 100 MHz s_memrealtime) while other contexts' kernels run.  Under the
 * verifier's load the chip holds 2.0-2.2 GHz, not the 2.4 GHz of a light kernel: issue-rate peaks have to be priced at
 * the clock measured (bench.py reports it as shader_clock_ghz). */
int bpp_shader_clock(bpp_ctx *ctx, uint32_t window_us, double *ghz);

/* diagnostics (tests/test_gpu_round3.py): number of non-zero bytes in the device copies of the secrets a resident batch
 * holds -- the statements' seed nonces and the recovered masks (zeroized on drop by the reference: src/range_statement.rs:76-81,
 * src/extended_mask.rs:14).  batch == 0 looks at the buffers the context kept from the last destroyed batch, which the
 * next upload will adopt: must read 0. */
int bpp_batch_secret_bytes(bpp_ctx *ctx, uint64_t batch, uint64_t *nonzero);
/* the same for the prover (tests/test_gpu_round5.py): the context's prover arena on the device -- witness bytes, bit vectors, nonces,
 * blinding-factor accumulators, the transcript-RNG states keyed with the witness -- and its page-locked staging on the way IN
 * (the witness bytes; on the way out only proofs and status words travel), which bpp_prove_batch wipes on EVERY exit path (the reference
 * keeps all of it in Zeroizing<>: src/range_proof.rs:300-301,325,438-464,542-571).  *examined = bytes looked at (0 before the
 * first prove call), *nonzero = how many of them are not zero: must read 0 between calls.  Not reachable from the host and
 * therefore not counted here: the LDS of the prover's kernels (generator states keyed with the witness, raw draws, digits of
 * witness-derived scalars) -- every such kernel clears its LDS as its last statement (kernels_prove.h: lds_wipe; ct.h). */
int bpp_prove_secret_bytes(bpp_ctx *ctx, uint64_t *examined, uint64_t *nonzero);

/* Transcript::new(label) -> 203-byte STROBE state (host helper for callers that keep merlin on their side) */
int bpp_transcript_new(const uint8_t *label, size_t label_len, uint8_t state203[203]);

#ifdef __cplusplus
}
#endif
#endif /* BPP_H */
