// Included at the end of src/range_proof.rs under `--features gpu` (see README.md next to this file): GPU forms of
// RangeProof::verify_batch (src/range_proof.rs:712-752) and RangeProof::prove_with_rng (:232-608) on libbpp_hip.so.
// Lives inside the crate because it reads private state: RangeProof's fields (a, a1, b, r1, s1, d1, li, ri: :58-68),
// RangeProofTranscript (pub(crate), src/transcripts.rs:36), NullRng (pub(crate), src/utils/nullrng.rs:16) and
// CommitmentOpening's `v` / `r` (pub(crate), src/commitment_opening.rs:14-20).  A child module of `range_proof` sees all of them.
//
// THREE entry points:
//
//   gpu::verify_batch(transcripts: &mut [Transcript], statements, proofs, action)
//       THE REFERENCE'S SIGNATURE AND SEMANTICS (:712-717).  merlin::Transcript has no state accessor, so the Fiat-Shamir
//       replay of PASS 1 (:811-850) stays in Rust, on the caller's own transcripts, through the crate's own
//       RangeProofTranscript: whatever the transcripts hold (any label, any context appended) binds the proofs exactly as in the
//       reference, and the transcripts come back advanced exactly as the reference leaves them (:757).  Everything after PASS 1
//       -- weight chain, decompression, the scalar block, mask recovery, the final multiscalar multiplication -- runs on the
//       engine (bpp_verify_batch_with_challenges, SURVEY 8b option (i)).  tests/ristretto.rs calls it unchanged.
//   gpu::verify_batch_fresh(GpuTranscript, statements, proofs, action)
//       the fast opt-in: the caller STATES what its transcripts are (a fresh Transcript::new(label), or a 203-byte STROBE
//       state) and PASS 1 runs on the device as well (one lane / one wavefront per proof instead of one host core for all).
//       Nothing is inferred: a caller that cannot say uses verify_batch above.
//   gpu::prove_with_rng(GpuTranscript, statement, witness, rng)
pub mod gpu {
    use alloc::vec::Vec;
    use core::convert::TryInto;  // (the crate is edition 2018: not in its prelude)
    use core::ops::{Add, Mul};

    use bpp_gpu_shim::{cached_params, default_engine, Action, Engine, GpuError, PackedBatch, Params, ProveItem, VerifyItem};
    use curve25519_dalek::{
        scalar::Scalar,
        traits::{Identity, IsIdentity, MultiscalarMul},
    };
    use merlin::Transcript;
    use rand_core::CryptoRngCore;
    use std::sync::Arc;
    use zeroize::Zeroizing;

    use super::{RangeProof, VerifyAction, MAX_RANGE_PROOF_BATCH_SIZE};
    use crate::{
        errors::ProofError,
        extended_mask::ExtendedMask,
        protocols::curve_point_protocol::CurvePointProtocol,
        range_statement::RangeStatement,
        range_witness::RangeWitness,
        traits::{Compressable, FixedBytesRepr, FromUniformBytes, Precomputable},
        transcripts::RangeProofTranscript,
        utils::nullrng::NullRng,
    };

    /// what the caller's transcripts are, for the entry points that run PASS 1 on the device (see the head of this file)
    #[derive(Clone, Copy)]
    pub enum GpuTranscript<'a> {
        Fresh(&'a [u8]),
        State(&'a [u8; 203]),
    }
    impl<'a> GpuTranscript<'a> {
        fn label(&self) -> &'a [u8] {
            match self {
                GpuTranscript::Fresh(l) => l,
                GpuTranscript::State(_) => &[],
            }
        }
        fn state(&self) -> Option<&'a [u8; 203]> {
            match self {
                GpuTranscript::Fresh(_) => None,
                GpuTranscript::State(s) => Some(s),
            }
        }
    }

    fn to_proof_error(e: GpuError) -> ProofError {
        match e {
            GpuError::VerificationFailed(m) => ProofError::VerificationFailed(m),
            GpuError::InvalidArgument(m) => ProofError::InvalidArgument(m),
            GpuError::InvalidLength(m) => ProofError::InvalidLength(m),
            GpuError::InvalidBlake2b => ProofError::InvalidBlake2b,
            GpuError::SizeOverflow => ProofError::SizeOverflow,
            GpuError::Engine(rc, m) => ProofError::InvalidArgument(alloc::format!("GPU engine fault {rc}: {m}")),
        }
    }

    fn to_action(action: VerifyAction) -> Action {
        match action {
            VerifyAction::VerifyOnly => Action::VerifyOnly,
            VerifyAction::RecoverAndVerify => Action::RecoverAndVerify,
            VerifyAction::RecoverOnly => Action::RecoverOnly,
        }
    }

    fn to_masks(degree: crate::generators::pedersen_gens::ExtensionDegree, masks: Vec<Option<Vec<[u8; 32]>>>) -> Vec<Option<ExtendedMask>> {
        masks.into_iter().map(|m| m.map(|b| {
            // (canonical by construction: the engine reduces mod l)
            let scalars: Vec<Scalar> = b.iter().map(|x| Scalar::from_canonical_bytes(*x).unwrap()).collect();
            ExtendedMask::assign(degree, scalars).unwrap()
        })).collect()
    }

    /// device tables of a statement's RangeParameters: built once per (bit length, aggregation, degree, bases) and process
    /// (bpp_gpu_shim::cached_params), shared by every call like the reference's Arc'd generators; nothing to release here
    fn params_for<P>(st: &RangeStatement<P>) -> Result<Arc<Params>, GpuError>
    where P: Compressable + FromUniformBytes + Clone + Precomputable, P::Compressed: FixedBytesRepr {
        let g = &st.generators;
        let gb: Vec<u8> = g.g_bases_compressed().iter().flat_map(|c| *c.as_fixed_bytes()).collect();
        cached_params(g.bit_length(), g.max_aggregation_factor(), g.extension_degree() as usize, g.h_base_compressed().as_fixed_bytes(), &gb)
    }

    /// one bpp_batcher per parameter set and process (round 4: the batcher pools every shape and every VerifyAction)
    fn pooled_batcher(params: &Arc<Params>) -> Result<Arc<bpp_gpu_shim::Batcher>, GpuError> {
        use std::collections::HashMap;
        use std::sync::{Mutex, OnceLock};
        static POOL: OnceLock<Mutex<HashMap<usize, Arc<bpp_gpu_shim::Batcher>>>> = OnceLock::new();
        let key = Arc::as_ptr(params) as usize;
        let mut pool = POOL.get_or_init(|| Mutex::new(HashMap::new())).lock().unwrap();
        if let Some(b) = pool.get(&key) {
            return Ok(b.clone());
        }
        // a context of its own for the batcher's first lane: the default engine stays free for the large calls
        let engine = Engine::new(0)?;
        let b = Arc::new(bpp_gpu_shim::Batcher::new_owning(engine, params, 0, 0, 0)?);
        pool.insert(key, b.clone());
        Ok(b)
    }

    // ------------------------------------------------------------------------------------------------------------------
    // the reference's own signature
    // ------------------------------------------------------------------------------------------------------------------

    /// `RangeProof::verify_batch` (src/range_proof.rs:712-752), signature and semantics: the proofs are bound to whatever the
    /// caller's transcripts hold, and the transcripts are advanced as `verify` advances them (:757, :811-850).
    /// Deviation kept from the engine: EVERY chunk of MAX_RANGE_PROOF_BATCH_SIZE proofs is verified, each as its own
    /// `verify` call with its own weight transcript, the first failing chunk's error is returned (the reference stops after the
    /// first chunk, :740-751, and never looks at the rest).
    pub fn verify_batch<P>(transcripts: &mut [Transcript], statements: &[RangeStatement<P>], proofs: &[RangeProof<P>], action: VerifyAction)
                           -> Result<Vec<Option<ExtendedMask>>, ProofError>
    where
        for<'p> &'p P: Mul<Scalar, Output = P>,
        for<'p> &'p P: Add<Output = P>,
        P: CurvePointProtocol + Precomputable + MultiscalarMul<Point = P>,
        P::Compressed: FixedBytesRepr + IsIdentity + Identity,
    {
        // :719-734, same order, same messages
        if statements.is_empty() || proofs.is_empty() || transcripts.is_empty() {
            return Err(ProofError::InvalidArgument("Range statements or proofs length empty".into()));
        }
        if statements.len() != proofs.len() {
            return Err(ProofError::InvalidArgument("Range statements and proofs length mismatch".into()));
        }
        if transcripts.len() != statements.len() {
            return Err(ProofError::InvalidArgument("Range statements and transcripts length mismatch".into()));
        }
        let mut masks = Vec::<Option<ExtendedMask>>::with_capacity(proofs.len());
        let chunks = statements.chunks(MAX_RANGE_PROOF_BATCH_SIZE).zip(proofs.chunks(MAX_RANGE_PROOF_BATCH_SIZE))
                               .zip(transcripts.chunks_mut(MAX_RANGE_PROOF_BATCH_SIZE));
        for ((batch_statements, batch_proofs), batch_transcripts) in chunks {
            let mut result = verify_chunk(batch_transcripts, batch_statements, batch_proofs, action)?;
            masks.append(&mut result);
        }
        Ok(masks)
    }

    /// one `RangeProof::verify` call (:756-1065): consistency loops and PASS 1 here, the arithmetic on the engine
    fn verify_chunk<P>(transcripts: &mut [Transcript], statements: &[RangeStatement<P>], proofs: &[RangeProof<P>], action: VerifyAction)
                       -> Result<Vec<Option<ExtendedMask>>, ProofError>
    where
        for<'p> &'p P: Mul<Scalar, Output = P>,
        for<'p> &'p P: Add<Output = P>,
        P: CurvePointProtocol + Precomputable + MultiscalarMul<Point = P>,
        P::Compressed: FixedBytesRepr + IsIdentity + Identity,
    {
        // (1) :766 -- the reference's own consistency loops come first, so a finding of theirs precedes any PASS-1 finding
        let (_max_mn, max_index) = RangeProof::<P>::verify_statements_and_generators_consistency(statements, proofs)?;
        let first_statement = statements.first().ok_or(ProofError::InvalidArgument("Empty proof statements".into()))?;
        let max_statement = statements.get(max_index).ok_or(ProofError::InvalidArgument("Out of bounds statement index".into()))?;
        let bit_length = first_statement.generators.bit_length();
        let extension_degree = first_statement.generators.extension_degree() as usize;
        let g_bases_compressed = first_statement.generators.g_bases_compressed();
        let h_base_compressed = first_statement.generators.h_base_compressed();

        // (2) :811-850 -- PASS 1 on the caller's transcripts, line for line: RangeProofTranscript::new, y / z, the round
        // challenges, the final challenge, to_verifier_rng, 32 bytes.  An error returns at once, as `?` does in the reference:
        // the transcripts of the proofs before it have been advanced, the later ones have not.
        let n = proofs.len();
        let mut challenges: Vec<Vec<u8>> = Vec::with_capacity(n);   // per proof: y, z, e_0.., e_final, 32 bytes each
        let mut rng_out: Vec<u8> = Vec::with_capacity(32 * n);      // what the reference appends to the weight transcript (:849)
        for ((proof, statement), transcript) in proofs.iter().zip(statements.iter()).zip(transcripts.iter_mut()) {
            let mut null_rng = NullRng;
            let mut transcript = RangeProofTranscript::<P, NullRng>::new(
                transcript,
                &h_base_compressed,
                g_bases_compressed,
                bit_length,
                extension_degree,
                statement.commitments.len(),
                statement,
                None,
                &mut null_rng,
            )?;
            let (y, z) = transcript.challenges_y_z(&proof.a)?;
            let mut c = Vec::with_capacity(32 * (proof.li.len() + 3));
            c.extend_from_slice(y.as_bytes());
            c.extend_from_slice(z.as_bytes());
            for (l, r) in proof.li.iter().zip(proof.ri.iter()) {
                c.extend_from_slice(transcript.challenge_round_e(l, r)?.as_bytes());
            }
            c.extend_from_slice(transcript.challenge_final_e(&proof.a1, &proof.b)?.as_bytes());
            let mut transcript_rng = transcript.to_verifier_rng(&proof.r1, &proof.s1, &proof.d1);
            let mut bytes = [0u8; 32];
            let transcript_rng = transcript_rng.as_rngcore();
            transcript_rng.fill_bytes(&mut bytes);
            rng_out.extend_from_slice(&bytes);
            challenges.push(c);
        }

        // (3) everything else of verify() on the engine: the weight transcript over rng_out (:853, :894), decompression
        // (:859-866), the L/R count (:875-888), the scalar block (:894-1033), masks (:941-969), the final check (:1050-1062)
        let blobs: Vec<Vec<u8>> = proofs.iter().map(|p| p.to_bytes()).collect();
        let comms: Vec<Vec<u8>> = statements.iter().map(|s| s.commitments_compressed.iter().flat_map(|c| *c.as_fixed_bytes()).collect()).collect();
        // seed nonces are secrets (Zeroize for RangeStatement, src/range_statement.rs:76-81)
        let seeds: Zeroizing<Vec<u8>> = Zeroizing::new(statements.iter().flat_map(|s| s.seed_nonce.map_or([0u8; 32], |x| x.to_bytes())).collect());
        let seed_refs: Vec<Option<&[u8; 32]>> = (0..n).map(|i| {
            if statements[i].seed_nonce.is_some() { Some((&seeds[32 * i..32 * i + 32]).try_into().unwrap()) } else { None }
        }).collect();
        let items: Vec<VerifyItem<'_>> = (0..n).map(|i| VerifyItem {
            proof: &blobs[i][..], commitments: &comms[i][..], min_values: &statements[i].minimum_value_promises[..], seed_nonce: seed_refs[i],
            transcript_label: &[], transcript_state: None }).collect();   // (the transcript fields are ignored by this entry)
        let run = || -> Result<Vec<Option<Vec<[u8; 32]>>>, GpuError> {
            let params = params_for(max_statement)?;
            let engine = default_engine().lock().unwrap();
            // chunk = 0: these <= 256 proofs are ONE verify() call
            engine.verify_batch_with_challenges(&params, &items, &challenges, &rng_out, to_action(action), 0)
        };
        run().map_err(to_proof_error).map(|m| to_masks(max_statement.generators.extension_degree(), m))
    }

    // ------------------------------------------------------------------------------------------------------------------
    // the fast opt-in: the caller says what its transcripts are, PASS 1 runs on the device too
    // ------------------------------------------------------------------------------------------------------------------

    /// `RangeProof::verify_batch` for transcripts the caller DESCRIBES (every one of them is exactly `Transcript::new(label)`, or
    /// in the given STROBE state).  The caller's `Transcript` objects, if it has any, are not advanced.  Verifies EVERY chunk of
    /// MAX_RANGE_PROOF_BATCH_SIZE proofs (the reference stops after the first, :740-751).
    pub fn verify_batch_fresh<P>(transcripts: GpuTranscript<'_>, statements: &[RangeStatement<P>], proofs: &[RangeProof<P>], action: VerifyAction)
                                 -> Result<Vec<Option<ExtendedMask>>, ProofError>
    where P: Compressable + FromUniformBytes + Clone + Precomputable, P::Compressed: FixedBytesRepr {
        // :719-734
        if statements.is_empty() || proofs.is_empty() {
            return Err(ProofError::InvalidArgument("Range statements or proofs length empty".into()));
        }
        if statements.len() != proofs.len() {
            return Err(ProofError::InvalidArgument("Range statements and proofs length mismatch".into()));
        }
        let max = statements.iter().max_by_key(|s| s.generators.max_aggregation_factor()).unwrap();
        let act = to_action(action);
        let n = proofs.len();
        let m0 = statements[0].commitments_compressed.len();
        let blobs: Vec<Vec<u8>> = proofs.iter().map(|p| p.to_bytes()).collect();
        let homogeneous = statements.iter().all(|s| s.commitments_compressed.len() == m0) && blobs.iter().all(|b| b.len() == blobs[0].len());
        // seed nonces are secrets (Zeroize for RangeStatement, src/range_statement.rs:76-81)
        let seeds: Zeroizing<Vec<u8>> = Zeroizing::new(statements.iter().flat_map(|s| s.seed_nonce.map_or([0u8; 32], |x| x.to_bytes())).collect());
        let seed_present: Vec<u8> = statements.iter().map(|s| s.seed_nonce.is_some() as u8).collect();
        let run = || -> Result<Vec<Option<Vec<[u8; 32]>>>, GpuError> {
            let params = params_for(max)?;
            if homogeneous {
                let flat: Vec<u8> = blobs.concat();
                let comms: Vec<u8> = statements.iter().flat_map(|s| s.commitments_compressed.iter().flat_map(|c| *c.as_fixed_bytes())).collect();
                let mins: Vec<u64> = statements.iter().flat_map(|s| s.minimum_value_promises.iter().map(|v| v.unwrap_or(0))).collect();
                let present: Vec<u8> = statements.iter().flat_map(|s| s.minimum_value_promises.iter().map(|v| v.is_some() as u8)).collect();
                let input = PackedBatch { n_items: n, proofs: &flat[..], proof_len: blobs[0].len(), commitments: &comms[..], m: m0, min_values: &mins[..],
                                          min_present: &present[..], seed_nonces: Some((&seeds[..], &seed_present[..])),
                                          transcript_label: transcripts.label(), transcript_state: transcripts.state() };
                if n <= MAX_RANGE_PROOF_BATCH_SIZE {
                    // ONE reference batch: a small call.  Separate threads with such calls stop at about 5 000 calls per second
                    // on a context each (and at one call at a time behind the default engine's mutex); pooled with the other
                    // threads' calls by a process-wide bpp_batcher they share grouped engine calls -- any VerifyAction, any shape
                    // (bpp_batcher_verify_action).  Same outcome, same masks.
                    return pooled_batcher(&params)?.verify_action(&input, act, params.extension_degree());
                }
                let engine = default_engine().lock().unwrap();
                engine.verify_batch_packed(&params, &input, act, MAX_RANGE_PROOF_BATCH_SIZE)
            } else {
                let comms: Vec<Vec<u8>> = statements.iter().map(|s| s.commitments_compressed.iter().flat_map(|c| *c.as_fixed_bytes()).collect()).collect();
                let seed_refs: Vec<Option<&[u8; 32]>> = (0..n).map(|i| if seed_present[i] != 0 { Some((&seeds[32 * i..32 * i + 32]).try_into().unwrap()) } else { None }).collect();
                let items: Vec<VerifyItem<'_>> = (0..n).map(|i| VerifyItem {
                    proof: &blobs[i][..], commitments: &comms[i][..], min_values: &statements[i].minimum_value_promises[..],
                    seed_nonce: seed_refs[i], transcript_label: transcripts.label(), transcript_state: transcripts.state() }).collect();
                let engine = default_engine().lock().unwrap();
                engine.verify_batch(&params, &items, act, MAX_RANGE_PROOF_BATCH_SIZE)
            }
        };
        run().map_err(to_proof_error).map(|m| to_masks(max.generators.extension_degree(), m))
    }

    /// `RangeProof::prove_with_rng` (src/range_proof.rs:232-608) on the GPU for a transcript the caller describes
    pub fn prove_with_rng<P, R: CryptoRngCore>(transcript: GpuTranscript<'_>, statement: &RangeStatement<P>, witness: &RangeWitness, rng: &mut R)
                                               -> Result<RangeProof<P>, ProofError>
    where P: Compressable + FromUniformBytes + Clone + Precomputable, P::Compressed: FixedBytesRepr, RangeProof<P>: Sized {
        let m = statement.commitments.len();
        // :248-260
        if witness.openings.len() != m || m == 0 {
            return Err(ProofError::InvalidLength("Witness openings statement commitments do not match!".into()));
        }
        let rounds = (m * statement.generators.bit_length()).trailing_zeros() as usize;
        // exactly the draws TranscriptRngBuilder::finalize would make, in order: (rounds + 3) x 32 bytes.  They key the
        // blinding draws: wiped like the reference wipes what it derives from them (:300-301,325,438-464)
        let mut rng_bytes = Zeroizing::new(alloc::vec![0u8; 32 * (rounds + 3)]);
        for chunk in rng_bytes.chunks_mut(32) {
            rng.fill_bytes(chunk);
        }
        let values: Zeroizing<Vec<u64>> = Zeroizing::new(witness.openings.iter().map(|o| o.v).collect());
        let blindings: Zeroizing<Vec<u8>> = Zeroizing::new(witness.openings.iter().flat_map(|o| o.r.iter().flat_map(|s| s.to_bytes())).collect());
        let comms: Vec<u8> = statement.commitments_compressed.iter().flat_map(|c| *c.as_fixed_bytes()).collect();
        let seed: Zeroizing<Option<[u8; 32]>> = Zeroizing::new(statement.seed_nonce.map(|x| x.to_bytes()));
        let run = || -> Result<Vec<u8>, GpuError> {
            let params = params_for(statement)?;
            let engine = default_engine().lock().unwrap();
            let item = ProveItem { values: &values[..], blindings: &blindings[..], commitments: &comms[..],
                                   min_values: &statement.minimum_value_promises[..], seed_nonce: (*seed).as_ref(),
                                   transcript_label: transcript.label(), transcript_state: transcript.state(), rng_bytes: &rng_bytes[..] };
            let mut out = engine.prove_batch(&params, core::slice::from_ref(&item))?;
            Ok(out.remove(0))
        };
        run().map_err(to_proof_error).and_then(|bytes| RangeProof::<P>::from_bytes(&bytes))
    }
}
