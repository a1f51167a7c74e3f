//! Golden vectors from the UNPATCHED reference (tari_bulletproofs_plus 0.4.1) for tests/test_ref_golden.py.
//!
//! Shapes: the four integration tests of tests/ristretto.rs:24-142 (bit lengths, aggregation factors, extension degrees,
//! minimum-value strategies), built exactly like `prove_and_verify` (tests/ristretto.rs:152-227) builds them, plus one
//! 64-bit batch of 8 and the benches' recipe (benches/range_proof.rs:206-262).  For every case the file holds the inputs
//! (values, blindings, promises, seed nonce, label), the bytes `prove_with_rng` drew from the external RNG (recorded by
//! `RecordingRng`; the engine's prover takes exactly these), the commitments, the proof bytes, and what
//! `verify_batch` returns for the three `VerifyAction`s, for a wrong seed nonce and for a bumped promise.
use std::{env, fs};

use curve25519_dalek::{ristretto::RistrettoPoint, scalar::Scalar};
use merlin::Transcript;
use rand_chacha::ChaCha12Rng;
use rand_core::{CryptoRng, RngCore, SeedableRng};
use serde_json::{json, Value};
use tari_bulletproofs_plus::{
    commitment_opening::CommitmentOpening,
    errors::ProofError,
    extended_mask::ExtendedMask,
    generators::pedersen_gens::ExtensionDegree,
    protocols::scalar_protocol::ScalarProtocol,
    range_parameters::RangeParameters,
    range_proof::VerifyAction,
    range_statement::RangeStatement,
    range_witness::RangeWitness,
    ristretto,
    ristretto::RistrettoRangeProof,
    traits::FixedBytesRepr,
};

/// Hands out the inner RNG's bytes and keeps a copy of everything it handed out.
struct RecordingRng<'a, R: RngCore + CryptoRng> {
    inner: &'a mut R,
    log: Vec<u8>,
}
impl<'a, R: RngCore + CryptoRng> RngCore for RecordingRng<'a, R> {
    fn next_u32(&mut self) -> u32 {
        let mut b = [0u8; 4];
        self.fill_bytes(&mut b);
        u32::from_le_bytes(b)
    }
    fn next_u64(&mut self) -> u64 {
        let mut b = [0u8; 8];
        self.fill_bytes(&mut b);
        u64::from_le_bytes(b)
    }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        self.inner.fill_bytes(dest);
        self.log.extend_from_slice(dest);
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), rand_core::Error> {
        self.fill_bytes(dest);
        Ok(())
    }
}
impl<'a, R: RngCore + CryptoRng> CryptoRng for RecordingRng<'a, R> {}

fn masks_json(m: &[Option<ExtendedMask>]) -> Value {
    Value::Array(
        m.iter()
            .map(|x| match x {
                None => Value::Null,
                Some(mask) => json!(mask.blindings().unwrap().iter().map(|s| hex::encode(s.as_bytes())).collect::<Vec<_>>()),
            })
            .collect(),
    )
}

fn result_json(r: Result<Vec<Option<ExtendedMask>>, ProofError>) -> Value {
    match r {
        Ok(m) => json!({ "ok": masks_json(&m) }),
        Err(e) => json!({ "err": match e {
            ProofError::VerificationFailed(_) => 1,
            ProofError::InvalidArgument(_) => 2,
            ProofError::InvalidLength(_) => 3,
            ProofError::InvalidBlake2b => 4,
            ProofError::SizeOverflow => 5,
        }, "msg": e.to_string() }),
    }
}

#[derive(Clone, Copy)]
enum Strategy {
    NoOffset,
    Intermediate,
    EqualToValue,
}

fn case(name: &str, bit_length: usize, proof_batch: &[usize], degree: ExtensionDegree, strategy: Strategy, seed: u64) -> Value {
    let mut rng = ChaCha12Rng::seed_from_u64(seed);
    let label = "BatchedRangeProofTest";
    let value_max = (1u128 << (bit_length - 1)) as u64;
    let t = degree as usize;
    let (mut st_private, mut st_public, mut proofs, mut transcripts, mut items) = (vec![], vec![], vec![], vec![], vec![]);
    for &m in proof_batch {
        let generators = RangeParameters::init(bit_length, m, ristretto::create_pedersen_gens_with_extension_degree(degree)).unwrap();
        let (mut openings, mut commitments, mut mins, mut vals, mut blinds) = (vec![], vec![], vec![], vec![], vec![]);
        for _ in 0..m {
            let value = rng.next_u64() % value_max;
            mins.push(match strategy {
                Strategy::NoOffset => None,
                Strategy::Intermediate => Some(value / 3),
                Strategy::EqualToValue => Some(value),
            });
            let blindings = vec![Scalar::random_not_zero(&mut rng); t];
            commitments.push(generators.pc_gens().commit(&Scalar::from(value), blindings.as_slice()).unwrap());
            vals.push(value);
            blinds.push(blindings.iter().map(|s| hex::encode(s.as_bytes())).collect::<Vec<_>>());
            openings.push(CommitmentOpening::new(value, blindings));
        }
        let witness = RangeWitness::init(openings).unwrap();
        let seed_nonce = if m == 1 { Some(Scalar::random_not_zero(&mut rng)) } else { None };
        let private = RangeStatement::init(generators.clone(), commitments.clone(), mins.clone(), seed_nonce).unwrap();
        let public = RangeStatement::init(generators.clone(), commitments, mins.clone(), None).unwrap();
        let transcript = Transcript::new(label.as_bytes());
        let mut rec = RecordingRng { inner: &mut rng, log: vec![] };
        let proof = RistrettoRangeProof::prove_with_rng(&mut transcript.clone(), &private, &witness, &mut rec).unwrap();
        items.push(json!({
            "m": m, "values": vals, "blindings": blinds, "min_values": mins,
            "seed_nonce": seed_nonce.map(|s| hex::encode(s.as_bytes())),
            "commitments": private.commitments_compressed.iter().map(|c| hex::encode(c.as_fixed_bytes())).collect::<Vec<_>>(),
            "rng_bytes": hex::encode(&rec.log),
            "proof": hex::encode(proof.to_bytes()),
        }));
        st_private.push(private);
        st_public.push(public);
        proofs.push(proof);
        transcripts.push(transcript);
    }
    let v = |sts: &[RangeStatement<RistrettoPoint>], action: VerifyAction| result_json(RistrettoRangeProof::verify_batch(&mut transcripts.clone(), sts, &proofs, action));
    // wrong seed nonce (tests/ristretto.rs:291-318) and bumped promises (:320-352)
    let wrong_seed: Vec<_> = st_private.iter().map(|s| RangeStatement {
        generators: s.generators.clone(), commitments: s.commitments.clone(), commitments_compressed: s.commitments_compressed.clone(),
        minimum_value_promises: s.minimum_value_promises.clone(), seed_nonce: s.seed_nonce.map(|x| x + Scalar::ONE) }).collect();
    let bumped: Vec<_> = st_public.iter().map(|s| RangeStatement {
        generators: s.generators.clone(), commitments: s.commitments.clone(), commitments_compressed: s.commitments_compressed.clone(),
        minimum_value_promises: s.minimum_value_promises.iter().map(|p| Some(p.map_or(1, |v| v.saturating_add(1)))).collect(),
        seed_nonce: s.seed_nonce }).collect();
    json!({
        "name": name, "bit_length": bit_length, "aggregation": proof_batch, "extension_degree": t, "label": label, "rng_seed": seed,
        "items": items,
        "verify": {
            "private_recover_only": v(&st_private, VerifyAction::RecoverOnly),
            "private_recover_and_verify": v(&st_private, VerifyAction::RecoverAndVerify),
            "private_verify_only": v(&st_private, VerifyAction::VerifyOnly),
            "public_verify_only": v(&st_public, VerifyAction::VerifyOnly),
            "wrong_seed_recover_and_verify": v(&wrong_seed, VerifyAction::RecoverAndVerify),
            "bumped_promise_verify_only": v(&bumped, VerifyAction::VerifyOnly),
        },
    })
}

fn main() {
    let out = env::args().nth(1).unwrap_or_else(|| "ref_vectors.json".to_string());
    let degrees = [
        (ExtensionDegree::DefaultPedersen, Strategy::NoOffset),
        (ExtensionDegree::AddOneBasePoint, Strategy::Intermediate),
        (ExtensionDegree::AddTwoBasePoints, Strategy::EqualToValue),
    ];
    let mut cases = vec![];
    for (d, s) in degrees {
        let t = d as usize;
        for n in [8usize, 64] {
            cases.push(case(&format!("single_n{n}_t{t}"), n, &[1], d, s, 8675309)); // tests/ristretto.rs:24-53
        }
        for n in [4usize, 32] {
            cases.push(case(&format!("aggregated4_n{n}_t{t}"), n, &[4], d, s, 8675309)); // :55-84
        }
        cases.push(case(&format!("two_singles_n64_t{t}"), 64, &[1, 1], d, s, 8675309)); // :86-115
        cases.push(case(&format!("mixed_1_2_n64_t{t}"), 64, &[1, 2], d, s, 8675309)); // :117-142
    }
    cases.push(case("bench_recipe_8x1_n64_t1", 64, &[1; 8], ExtensionDegree::DefaultPedersen, Strategy::Intermediate, 8675309));
    cases.push(case("bench_recipe_2x8_n64_t1", 64, &[8, 8], ExtensionDegree::DefaultPedersen, Strategy::Intermediate, 8675309));
    // public anchors of the parameter derivation (src/ristretto.rs:67-112, src/generators/bulletproof_gens.rs:83-112)
    let p = RangeParameters::init(64, 2, ristretto::create_pedersen_gens_with_extension_degree(ExtensionDegree::AddFiveBasePoints)).unwrap();
    let anchors = json!({
        "h_base": hex::encode(p.h_base_compressed().as_fixed_bytes()),
        "g_bases": p.g_bases_compressed().iter().map(|g| hex::encode(g.as_fixed_bytes())).collect::<Vec<_>>(),
        "gi": p.gi_base_iter().map(|g| hex::encode(g.compress().as_bytes())).collect::<Vec<_>>(),
        "hi": p.hi_base_iter().map(|g| hex::encode(g.compress().as_bytes())).collect::<Vec<_>>(),
    });
    let doc = json!({ "source": "tari_bulletproofs_plus 0.4.1 (unpatched), rust/ref-dump", "cases": cases, "anchors_n64_m2_t6": anchors });
    fs::write(&out, serde_json::to_string_pretty(&doc).unwrap()).unwrap();
    eprintln!("wrote {out}");
}
