// Included at the end of src/range_proof.rs under `--features gpu` (see README.md next to this file): the bodies of
// RangeProof::verify_batch (src/range_proof.rs:712-752) and RangeProof::prove_with_rng (:232-608) on libbpp_hip.so.
// Lives inside the crate because it reads private state: RangeProof's fields through to_bytes(), CommitmentOpening's
// `v` / `r` (pub(crate), src/commitment_opening.rs:14-20).
pub(crate) mod gpu {
    use alloc::{string::String, vec::Vec};
    use std::sync::Mutex;

    use bpp_gpu_shim::{default_engine, Action, GpuError, Params, ProveItem, VerifyItem};
    use curve25519_dalek::scalar::Scalar;
    use merlin::Transcript;
    use rand_core::CryptoRngCore;

    use super::{RangeProof, VerifyAction, MAX_RANGE_PROOF_BATCH_SIZE};
    use crate::{
        errors::ProofError,
        extended_mask::ExtendedMask,
        range_statement::RangeStatement,
        range_witness::RangeWitness,
        traits::{Compressable, FixedBytesRepr, FromUniformBytes, Precomputable},
    };

    /// merlin::Transcript has no state accessor: callers tell the shim which label their (fresh) transcripts carry.
    static LABEL: Mutex<Option<Vec<u8>>> = Mutex::new(None);
    pub fn register_label(label: &[u8]) {
        *LABEL.lock().unwrap() = Some(label.to_vec());
    }
    pub fn label_of(_transcripts: &[Transcript]) -> Option<Vec<u8>> {
        LABEL.lock().unwrap().clone()
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

    /// device tables per (bit length, aggregation, degree, bases): created once, shared by every call (Arc semantics)
    fn params_for<P>(st: &RangeStatement<P>) -> Result<Params, GpuError>
    where P: Compressable + FromUniformBytes + Clone + Precomputable, P::Compressed: FixedBytesRepr {
        let g = &st.generators;
        let gb: Vec<u8> = g.g_bases_compressed().iter().flat_map(|c| *c.as_fixed_bytes()).collect();
        default_engine().lock().unwrap().params(g.bit_length(), g.max_aggregation_factor(), g.extension_degree() as usize,
                                                Some(g.h_base_compressed().as_fixed_bytes()), Some(&gb))
    }

    pub fn verify_batch<P>(statements: &[RangeStatement<P>], proofs: &[RangeProof<P>], action: VerifyAction, label: Option<Vec<u8>>)
                           -> Option<Result<Vec<Option<ExtendedMask>>, ProofError>>
    where P: Compressable + FromUniformBytes + Clone + Precomputable, P::Compressed: FixedBytesRepr {
        let label = label?;
        if statements.is_empty() || statements.len() != proofs.len() {
            return None; // the CPU path reports the argument errors of :719-734
        }
        let max = statements.iter().max_by_key(|s| s.generators.max_aggregation_factor())?;
        let run = || -> Result<Vec<Option<ExtendedMask>>, GpuError> {
            let params = params_for(max)?;
            let blobs: Vec<Vec<u8>> = proofs.iter().map(|p| p.to_bytes()).collect();
            let comms: Vec<Vec<u8>> = statements.iter().map(|s| s.commitments_compressed.iter().flat_map(|c| *c.as_fixed_bytes()).collect()).collect();
            let seeds: Vec<Option<[u8; 32]>> = statements.iter().map(|s| s.seed_nonce.map(|x| x.to_bytes())).collect();
            let items: Vec<VerifyItem<'_>> = (0..proofs.len()).map(|i| VerifyItem {
                proof: &blobs[i], commitments: &comms[i], min_values: &statements[i].minimum_value_promises,
                seed_nonce: seeds[i].as_ref(), transcript_label: &label, transcript_state: None }).collect();
            let act = match action {
                VerifyAction::VerifyOnly => Action::VerifyOnly,
                VerifyAction::RecoverAndVerify => Action::RecoverAndVerify,
                VerifyAction::RecoverOnly => Action::RecoverOnly,
            };
            let engine = default_engine().lock().unwrap();
            // the reference verifies the first MAX_RANGE_PROOF_BATCH_SIZE proofs only (:740-751); the engine verifies them all
            let masks = engine.verify_batch(&params, &items, act, MAX_RANGE_PROOF_BATCH_SIZE)?;
            let degree = max.generators.extension_degree();
            let out = masks.into_iter().map(|m| m.map(|b| {
                let scalars: Vec<Scalar> = b.iter().map(|x| Scalar::from_canonical_bytes(*x).unwrap()).collect();
                ExtendedMask::assign(degree, scalars).unwrap()
            })).collect();
            engine.release(params);
            Ok(out)
        };
        Some(run().map_err(to_proof_error))
    }

    pub fn prove_with_rng<P, R: CryptoRngCore>(statement: &RangeStatement<P>, witness: &RangeWitness, rng: &mut R, label: Option<Vec<u8>>)
                                               -> Option<Result<RangeProof<P>, ProofError>>
    where P: Compressable + FromUniformBytes + Clone + Precomputable, P::Compressed: FixedBytesRepr, RangeProof<P>: Sized {
        let label = label?;
        let m = statement.commitments.len();
        if witness.openings.len() != m || m == 0 {
            return None; // CPU path reports :248-260
        }
        let rounds = (m * statement.generators.bit_length()).trailing_zeros() as usize;
        // exactly the draws TranscriptRngBuilder::finalize would make, in order: (rounds + 3) x 32 bytes
        let mut rng_bytes = alloc::vec![0u8; 32 * (rounds + 3)];
        for chunk in rng_bytes.chunks_mut(32) {
            rng.fill_bytes(chunk);
        }
        let values: Vec<u64> = witness.openings.iter().map(|o| o.v).collect();
        let blindings: Vec<u8> = witness.openings.iter().flat_map(|o| o.r.iter().flat_map(|s| s.to_bytes())).collect();
        let comms: Vec<u8> = statement.commitments_compressed.iter().flat_map(|c| *c.as_fixed_bytes()).collect();
        let seed = statement.seed_nonce.map(|x| x.to_bytes());
        let run = || -> Result<Vec<u8>, GpuError> {
            let params = params_for(statement)?;
            let engine = default_engine().lock().unwrap();
            let item = ProveItem { values: &values, blindings: &blindings, commitments: &comms, min_values: &statement.minimum_value_promises,
                                   seed_nonce: seed.as_ref(), transcript_label: &label, transcript_state: None, rng_bytes: &rng_bytes };
            let mut out = engine.prove_batch(&params, core::slice::from_ref(&item))?;
            engine.release(params);
            Ok(out.remove(0))
        };
        Some(run().map_err(to_proof_error).and_then(|bytes| RangeProof::<P>::from_bytes(&bytes)))
    }

    #[allow(dead_code)]
    fn _unused(_: String) {}
}
