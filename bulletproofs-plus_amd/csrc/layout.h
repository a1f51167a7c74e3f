// Plain-data descriptors shared by the host packer (upload_host.h, compiled by hipcc AND by g++ for the sanitizer
// harness) and the device kernels (kernels_verify.h).
#pragma once
#include <stdint.h>

namespace bpp {

struct ProofDesc {
  uint32_t proof_off;   // byte offset of the proof in bytes[]
  uint32_t rounds;      // number of (L,R) pairs present in the proof
  uint32_t m;           // aggregation factor of the statement
  uint32_t commit_off;  // byte offset of the m compressed commitments in bytes[]
  uint32_t minval_idx;  // index of the first minimum value
  uint32_t dyn_off;     // index of the first dynamic (scalar, point) slot
  uint32_t state_idx;   // which initial transcript state
  uint32_t flags;       // bit0: seed nonce present
};

// status bits written by the kernels
#define BPP_ST_TRANSCRIPT_FAIL 1u  // identity encoding appended or zero challenge -> VerificationFailed
#define BPP_ST_DECOMPRESS_FAIL 2u  // proof point not a canonical encoding          -> InvalidArgument
#define BPP_ST_COMMIT_FAIL 4u      // statement commitment does not decode           -> InvalidArgument

// The engine's own limit on the number of (L, R) pairs of one proof.  No statement can need more than 11 (mn <= 2048).
// The reference rejects 32 or more with SizeOverflow in PASS 2 (src/range_proof.rs:875-888) after replaying all of them in
// PASS 1; the engine keeps that precedence for proofs of up to this many pairs and refuses longer ones at upload with the
// same error kind.  Kept small on purpose: PASS 1 replays EVERY pair of a proof on one lane (or one wavefront), about three
// Keccak-f each, so a cap of 2^20 (the earlier value: a 64 MB proof) let ONE hostile item hold the stream -- and every call
// queued behind it on the chip -- for seconds inside a single kernel.  At 64 pairs (a 4.3 KB proof) that kernel stays in
// the tens of microseconds.
#define BPP_MAX_WIRE_ROUNDS 64u

// bytes[] ends with this much zeroed slack: an item whose extension degree differs from the parameters' is reported when
// its chunk is verified (before anything else of that chunk), but when the call has several chunks the kernels still run
// over it with the parameters' layout -- the other chunks' verdicts are needed -- and may then address up to 5 x 32 bytes
// past its own end.  What they compute for it is never looked at: its chunk has failed already.
#define BPP_BYTES_SLACK 256u

}  // namespace bpp
