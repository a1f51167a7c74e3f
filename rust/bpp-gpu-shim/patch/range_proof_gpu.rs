// Included at the end of src/range_proof.rs under `--features gpu` (see README.md next to this file): GPU forms of
// RangeProof::verify_batch (src/range_proof.rs:712-752) and RangeProof::prove_with_rng (:232-608) on libbpp_hip.so.
// Lives inside the crate because it reads private state: RangeProof's fields through to_bytes(), CommitmentOpening's
// `v` / `r` (pub(crate), src/commitment_opening.rs:14-20).
//
// The transcript is an EXPLICIT argument.  merlin::Transcript has no state accessor, so these entry points cannot learn
// what a caller's `&mut Transcript` holds; an earlier draft guessed it from a process-global registered label, which bound
// proofs to the wrong transcript as soon as a caller had appended context data or used another label, and raced between
// threads.  Here the caller states what its transcripts are:
//   GpuTranscript::Fresh(label)   every transcript is exactly `Transcript::new(label)`, nothing appended
//                                 (benches/range_proof.rs:98, tests/ristretto.rs:225)
//   GpuTranscript::State(bytes)   the 203-byte STROBE state of the callers' transcripts (for callers that keep their own
//                                 merlin fork / state dump)
// A caller that cannot say uses the unpatched CPU entry points, which stay as they are.
pub mod gpu {
    use alloc::vec::Vec;

    use bpp_gpu_shim::{cached_params, default_engine, Action, GpuError, PackedBatch, Params, ProveItem, VerifyItem};
    use curve25519_dalek::scalar::Scalar;
    use rand_core::CryptoRngCore;
    use std::sync::Arc;
    use zeroize::Zeroizing;

    use super::{RangeProof, VerifyAction, MAX_RANGE_PROOF_BATCH_SIZE};
    use crate::{
        errors::ProofError,
        extended_mask::ExtendedMask,
        range_statement::RangeStatement,
        range_witness::RangeWitness,
        traits::{Compressable, FixedBytesRepr, FromUniformBytes, Precomputable},
    };

    /// what the caller's transcripts are (see the head of this file)
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

    /// device tables of a statement's RangeParameters: built once per (bit length, aggregation, degree, bases) and process
    /// (bpp_gpu_shim::cached_params), shared by every call like the reference's Arc'd generators; nothing to release here
    fn params_for<P>(st: &RangeStatement<P>) -> Result<Arc<Params>, GpuError>
    where P: Compressable + FromUniformBytes + Clone + Precomputable, P::Compressed: FixedBytesRepr {
        let g = &st.generators;
        let gb: Vec<u8> = g.g_bases_compressed().iter().flat_map(|c| *c.as_fixed_bytes()).collect();
        cached_params(g.bit_length(), g.max_aggregation_factor(), g.extension_degree() as usize, g.h_base_compressed().as_fixed_bytes(), &gb)
    }

    /// one bpp_batcher per (parameter set, proof length, aggregation factor, transcript label) and process
    fn pooled_batcher(params: &Arc<Params>, shape: &PackedBatch<'_>) -> Result<Arc<bpp_gpu_shim::Batcher>, GpuError> {
        use std::collections::HashMap;
        use std::sync::{Mutex, OnceLock};
        static POOL: OnceLock<Mutex<HashMap<(usize, usize, usize, Vec<u8>), Arc<bpp_gpu_shim::Batcher>>>> = OnceLock::new();
        let key = (Arc::as_ptr(params) as usize, shape.proof_len, shape.m, shape.transcript_label.to_vec());
        let mut pool = POOL.get_or_init(|| Mutex::new(HashMap::new())).lock().unwrap();
        if let Some(b) = pool.get(&key) {
            return Ok(b.clone());
        }
        // a context of its own for the batcher's first lane: the default engine stays free for the large calls
        let engine = Engine::new(0)?;
        let b = Arc::new(bpp_gpu_shim::Batcher::new_owning(engine, params, shape, 0, 0, 0)?);
        pool.insert(key, b.clone());
        Ok(b)
    }

    /// `RangeProof::verify_batch` (src/range_proof.rs:712-752) on the GPU, for transcripts the caller describes.  Verifies
    /// EVERY chunk of MAX_RANGE_PROOF_BATCH_SIZE proofs (the reference stops after the first, :740-751).
    pub fn verify_batch<P>(transcripts: GpuTranscript<'_>, statements: &[RangeStatement<P>], proofs: &[RangeProof<P>], action: VerifyAction)
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
        let act = match action {
            VerifyAction::VerifyOnly => Action::VerifyOnly,
            VerifyAction::RecoverAndVerify => Action::RecoverAndVerify,
            VerifyAction::RecoverOnly => Action::RecoverOnly,
        };
        let n = proofs.len();
        let m0 = statements[0].commitments_compressed.len();
        let blobs: Vec<Vec<u8>> = proofs.iter().map(|p| p.to_bytes()).collect();
        let homogeneous = statements.iter().all(|s| s.commitments_compressed.len() == m0) && blobs.iter().all(|b| b.len() == blobs[0].len());
        // seed nonces are secrets (Zeroize for RangeStatement, src/range_statement.rs:76-81)
        let seeds: Zeroizing<Vec<u8>> = Zeroizing::new(statements.iter().flat_map(|s| s.seed_nonce.map_or([0u8; 32], |x| x.to_bytes())).collect());
        let seed_present: Vec<u8> = statements.iter().map(|s| s.seed_nonce.is_some() as u8).collect();
        let run = || -> Result<Vec<Option<Vec<[u8; 32]>>>, GpuError> {
            let params = params_for(max)?;
            let engine = default_engine().lock().unwrap();
            if homogeneous {
                let flat: Vec<u8> = blobs.concat();
                let comms: Vec<u8> = statements.iter().flat_map(|s| s.commitments_compressed.iter().flat_map(|c| *c.as_fixed_bytes())).collect();
                let mins: Vec<u64> = statements.iter().flat_map(|s| s.minimum_value_promises.iter().map(|v| v.unwrap_or(0))).collect();
                let present: Vec<u8> = statements.iter().flat_map(|s| s.minimum_value_promises.iter().map(|v| v.is_some() as u8)).collect();
                let input = PackedBatch { n_items: n, proofs: &flat, proof_len: blobs[0].len(), commitments: &comms, m: m0, min_values: &mins,
                                          min_present: &present, seed_nonces: Some((&seeds, &seed_present)),
                                          transcript_label: transcripts.label(), transcript_state: transcripts.state() };
                if n <= MAX_RANGE_PROOF_BATCH_SIZE && matches!(action, VerifyAction::VerifyOnly) && transcripts.state().is_none() {
                    // ONE reference batch and nothing to recover: a small call.  Separate threads with such calls stop at about
                    // 5 000 calls per second on a context each (and at one call at a time behind the engine mutex above); pooled
                    // with the other threads' calls by a process-wide bpp_batcher they share grouped engine calls.  Same outcome.
                    drop(engine);
                    return pooled_batcher(&params, &input)?.verify(&input).map(|_| (0..n).map(|_| None).collect());
                }
                engine.verify_batch_packed(&params, &input, act, MAX_RANGE_PROOF_BATCH_SIZE)
            } else {
                let comms: Vec<Vec<u8>> = statements.iter().map(|s| s.commitments_compressed.iter().flat_map(|c| *c.as_fixed_bytes()).collect()).collect();
                let seed_refs: Vec<Option<&[u8; 32]>> = (0..n).map(|i| if seed_present[i] != 0 { Some(seeds[32 * i..32 * i + 32].try_into().unwrap()) } else { None }).collect();
                let items: Vec<VerifyItem<'_>> = (0..n).map(|i| VerifyItem {
                    proof: &blobs[i], commitments: &comms[i], min_values: &statements[i].minimum_value_promises,
                    seed_nonce: seed_refs[i], transcript_label: transcripts.label(), transcript_state: transcripts.state() }).collect();
                engine.verify_batch(&params, &items, act, MAX_RANGE_PROOF_BATCH_SIZE)
            }
        };
        let degree = max.generators.extension_degree();
        run().map_err(to_proof_error).map(|masks| masks.into_iter().map(|m| m.map(|b| {
            let scalars: Vec<Scalar> = b.iter().map(|x| Scalar::from_canonical_bytes(*x).unwrap()).collect();
            ExtendedMask::assign(degree, scalars).unwrap()
        })).collect())
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
            let item = ProveItem { values: &values, blindings: &blindings, commitments: &comms, min_values: &statement.minimum_value_promises,
                                   seed_nonce: seed.as_ref(), transcript_label: transcript.label(), transcript_state: transcript.state(),
                                   rng_bytes: &rng_bytes };
            let mut out = engine.prove_batch(&params, core::slice::from_ref(&item))?;
            Ok(out.remove(0))
        };
        run().map_err(to_proof_error).and_then(|bytes| RangeProof::<P>::from_bytes(&bytes))
    }
}
