//! Seam B1 of the engine (include/bpp.h "B1: multiscalar traits") as code: `GpuRistretto`, a point type the
//! UNPATCHED reference crate can be instantiated with -- `RangeProof<GpuRistretto>` -- whose three dalek multiscalar traits
//! forward to `libbpp_hip.so`:
//!
//! | reference bound (src/range_proof.rs:207-213, src/traits.rs:40-43)            | here                       | C entry point        |
//! |-------------------------------------------------------------------------------|----------------------------|----------------------|
//! | `P::Precomputation: VartimePrecomputedMultiscalarMul` (`new`, `optional_mixed_multiscalar_mul`) | `GpuPrecomputation` | `bpp_precomp_create`, `bpp_msm_mixed` |
//! | `P: VartimeMultiscalarMul` (`optional_multiscalar_mul`; via `CurvePointProtocol`)              | `GpuRistretto`      | `bpp_msm_vartime`    |
//! | `P: MultiscalarMul` (`multiscalar_mul`; `PedersenGens::commit`, src/generators/pedersen_gens.rs:120) | `GpuRistretto` | none: dalek's CONSTANT-TIME implementation on the host (the scalars are the value and the blinding factors) |
//!
//! Everything else a backend owes the crate (src/ristretto.rs:28-64 shows the list for dalek's own point: `Identity`, `Add`,
//! `AddAssign`, `PartialEq`, `Clone`, `&P * Scalar`, `&P + &P`, `Compressable` / `Decompressable`, `FixedBytesRepr`,
//! `FromUniformBytes`, `CurvePointProtocol`) is group arithmetic on single points and stays on dalek: `GpuRistretto` is a newtype
//! over `RistrettoPoint`, `GpuCompressedRistretto` one over `CompressedRistretto`.  Points cross the C boundary as their
//! 32-byte encodings (dalek's in-memory `RistrettoPoint` is not a stable ABI).
//!
//! What B1 is for: strict drop-in with ZERO edits to the reference crate, and differential testing of the MSM kernels against
//! dalek's Straus / Pippenger on the reference's own call sites.  It cannot reach the throughput of the protocol-level seam
//! (`bpp_gpu_shim` + `patch/range_proof_gpu.rs`): hashing, decompression and the scalar block stay on one host core.
//!
//! NOT COMPILED in the build container (no cargo).  Desk-checked against curve25519-dalek 4.1.3 `src/traits.rs` from memory
//! [dalek-knowledge]: if `VartimePrecomputedMultiscalarMul` of the pinned dalek has no `len` / `is_empty` (they arrived in the
//! 4.1.x series), delete those two methods below; nothing else depends on them.
use core::borrow::Borrow;
use core::ops::{Add, AddAssign, Mul};
use std::sync::{Mutex, OnceLock};

use bpp_gpu_shim::{Engine, Precomp};
use curve25519_dalek::{
    ristretto::{CompressedRistretto, RistrettoPoint},
    scalar::Scalar,
    traits::{Identity, MultiscalarMul, VartimeMultiscalarMul, VartimePrecomputedMultiscalarMul},
};
use subtle::{Choice, ConstantTimeEq};
use zeroize::Zeroizing;
use tari_bulletproofs_plus::{
    protocols::curve_point_protocol::CurvePointProtocol,
    traits::{Compressable, Decompressable, FixedBytesRepr, FromUniformBytes, Precomputable},
};

/// one context for the B1 calls of this process (they are blocking and short; callers that want concurrency use seam B2)
fn engine() -> &'static Mutex<Engine> {
    static ENGINE: OnceLock<Mutex<Engine>> = OnceLock::new();
    ENGINE.get_or_init(|| Mutex::new(Engine::new(0).expect("bpp-gpu-ristretto: no usable gfx950 device (there is no CPU fallback)")))
}

/// A ristretto255 point whose multiscalar multiplications run on the MI355X.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct GpuRistretto(pub RistrettoPoint);

/// Its 32-byte encoding.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct GpuCompressedRistretto(pub CompressedRistretto);

// ---- single-point group arithmetic: dalek's
impl Identity for GpuRistretto {
    fn identity() -> Self {
        GpuRistretto(RistrettoPoint::identity())
    }
}
impl Add for GpuRistretto {
    type Output = GpuRistretto;
    fn add(self, rhs: GpuRistretto) -> GpuRistretto {
        GpuRistretto(self.0 + rhs.0)
    }
}
impl<'a, 'b> Add<&'b GpuRistretto> for &'a GpuRistretto {
    type Output = GpuRistretto;
    fn add(self, rhs: &'b GpuRistretto) -> GpuRistretto {
        GpuRistretto(self.0 + rhs.0)
    }
}
impl AddAssign for GpuRistretto {
    fn add_assign(&mut self, rhs: GpuRistretto) {
        self.0 += rhs.0;
    }
}
impl<'a> Mul<Scalar> for &'a GpuRistretto {
    type Output = GpuRistretto;
    fn mul(self, rhs: Scalar) -> GpuRistretto {
        GpuRistretto(self.0 * rhs)
    }
}

// ---- encodings (src/traits.rs:7-37, as src/ristretto.rs:30-60 does for dalek's own types)
impl FixedBytesRepr for GpuCompressedRistretto {
    fn as_fixed_bytes(&self) -> &[u8; 32] {
        self.0.as_bytes()
    }
    fn from_fixed_bytes(bytes: [u8; 32]) -> Self {
        GpuCompressedRistretto(CompressedRistretto(bytes))
    }
}
impl Decompressable for GpuCompressedRistretto {
    type Decompressed = GpuRistretto;
    fn decompress(&self) -> Option<GpuRistretto> {
        self.0.decompress().map(GpuRistretto)
    }
}
impl Compressable for GpuRistretto {
    type Compressed = GpuCompressedRistretto;
    fn compress(&self) -> GpuCompressedRistretto {
        GpuCompressedRistretto(self.0.compress())
    }
}
impl FromUniformBytes for GpuRistretto {
    fn from_uniform_bytes(bytes: &[u8; 64]) -> Self {
        GpuRistretto(RistrettoPoint::from_uniform_bytes(bytes))
    }
}
// `P::Compressed: FixedBytesRepr + IsIdentity + Identity` (src/range_proof.rs:212): dalek gives `IsIdentity` to every type that
// is `ConstantTimeEq + Identity` (blanket impl in its traits.rs), so those two are what is implemented here
impl Identity for GpuCompressedRistretto {
    fn identity() -> Self {
        GpuCompressedRistretto(CompressedRistretto::identity())
    }
}
impl ConstantTimeEq for GpuCompressedRistretto {
    fn ct_eq(&self, other: &Self) -> Choice {
        self.0.ct_eq(&other.0)
    }
}
impl CurvePointProtocol for GpuRistretto {}  // (hash_from_bytes_sha3_512 is provided: SHA3-512 -> from_uniform_bytes)

// ---- the two VARIABLE-TIME multiscalar traits: on the device.  The constant-time one stays on dalek (below).
/// The scalars of the variable-time calls are witness-derived in the prover (src/range_proof.rs:482-495: the reference keeps
/// them in `Zeroizing` vectors): the byte copy made for the C call is wiped when it goes out of scope.  What the engine holds
/// of them on its side (page-locked staging, device buffers of `bpp_msm_vartime`) is NOT wiped by that entry point -- a caller
/// who proves with secrets that matter uses seam B2 (`bpp_prove_batch` wipes everything witness-derived), B1 is for
/// verification and for differential testing.
fn scalar_bytes<I>(scalars: I) -> Zeroizing<Vec<u8>>
where I: IntoIterator, I::Item: Borrow<Scalar> {
    Zeroizing::new(scalars.into_iter().flat_map(|s| {
        let s: &Scalar = s.borrow();
        s.to_bytes()
    }).collect())
}
fn decode(out: [u8; 32]) -> GpuRistretto {
    // the engine returns a canonical encoding of the sum it computed
    GpuRistretto(CompressedRistretto(out).decompress().expect("libbpp_hip.so returned a non-canonical point"))
}

impl VartimeMultiscalarMul for GpuRistretto {
    type Point = GpuRistretto;

    /// src/range_proof.rs:482-495, :512-521 (through the provided `vartime_multiscalar_mul`)
    fn optional_multiscalar_mul<I, J>(scalars: I, points: J) -> Option<GpuRistretto>
    where
        I: IntoIterator,
        I::Item: Borrow<Scalar>,
        J: IntoIterator<Item = Option<GpuRistretto>>,
    {
        let s = scalar_bytes(scalars);
        let mut p = Vec::with_capacity(s.len());
        for q in points {
            p.extend_from_slice(q?.0.compress().as_bytes());   // a missing point: None, as dalek's implementations do
        }
        assert_eq!(s.len(), p.len(), "scalars and points differ in number");
        Some(decode(engine().lock().unwrap().msm_vartime(&s, &p).expect("bpp_msm_vartime")))
    }
}

impl MultiscalarMul for GpuRistretto {
    type Point = GpuRistretto;

    /// src/generators/pedersen_gens.rs:120 (`PedersenGens::commit`): the reference calls the CONSTANT-TIME trait here because
    /// the scalars are secrets and nothing else (the value and its blinding factors).  It therefore stays what it is in the
    /// reference: dalek's constant-time Straus on the host, on the wrapped `RistrettoPoint`s -- 1 + t terms, microseconds; a
    /// device round trip would be slower as well as variable-time.  (Many commitments at once, on the device, in the
    /// uniform-access form: `bpp_gpu_shim::Engine::pedersen_commit` / `bpp_pedersen_commit`, include/bpp.h.)
    fn multiscalar_mul<I, J>(scalars: I, points: J) -> GpuRistretto
    where
        I: IntoIterator,
        I::Item: Borrow<Scalar>,
        J: IntoIterator,
        J::Item: Borrow<GpuRistretto>,
    {
        GpuRistretto(RistrettoPoint::multiscalar_mul(scalars, points.into_iter().map(|q| q.borrow().0)))
    }
}

/// `VartimeRistrettoPrecomputation`'s stand-in (src/ristretto.rs:62-64): the generator table lives on the device
pub struct GpuPrecomputation {
    table: Precomp,
}

impl VartimePrecomputedMultiscalarMul for GpuPrecomputation {
    type Point = GpuRistretto;

    /// src/generators/bulletproof_gens.rs:99-103: the interleaved G_i, H_i of every party
    fn new<I>(static_points: I) -> Self
    where
        I: IntoIterator,
        I::Item: Borrow<GpuRistretto>,
    {
        let p: Vec<u8> = static_points.into_iter().flat_map(|q| {
            let q: &GpuRistretto = q.borrow();
            q.0.compress().to_bytes()
        }).collect();
        GpuPrecomputation { table: engine().lock().unwrap().precomp(&p).expect("bpp_precomp_create") }
    }

    fn len(&self) -> usize {
        self.table.len()
    }

    fn is_empty(&self) -> bool {
        self.table.is_empty()
    }

    /// src/range_proof.rs:339-345 (the prover's A) and :1050-1057 (the verifier's final check), through the provided
    /// `vartime_mixed_multiscalar_mul`.  Fewer static scalars than table entries = zero padding (src/utils/generic.rs:63-82).
    fn optional_mixed_multiscalar_mul<I, J, K>(&self, static_scalars: I, dynamic_scalars: J, dynamic_points: K) -> Option<GpuRistretto>
    where
        I: IntoIterator,
        I::Item: Borrow<Scalar>,
        J: IntoIterator,
        J::Item: Borrow<Scalar>,
        K: IntoIterator<Item = Option<GpuRistretto>>,
    {
        let ss = scalar_bytes(static_scalars);
        let ds = scalar_bytes(dynamic_scalars);
        let mut dp = Vec::with_capacity(ds.len());
        for q in dynamic_points {
            dp.extend_from_slice(q?.0.compress().as_bytes());
        }
        assert_eq!(ds.len(), dp.len(), "dynamic scalars and points differ in number");
        assert!(ss.len() / 32 <= self.table.len(), "more static scalars than precomputed points");
        Some(decode(engine().lock().unwrap().msm_mixed(&self.table, &ss, &ds, &dp).expect("bpp_msm_mixed")))
    }
}

impl Precomputable for GpuRistretto {
    type Precomputation = GpuPrecomputation;   // `Send + Sync`: bpp_gpu_shim::Precomp is (a process-wide, read-only device table)
}

/// `create_pedersen_gens_with_extension_degree` (src/ristretto.rs:67-76) for this point type: the same bases, wrapped
pub fn create_pedersen_gens_with_extension_degree(
    extension_degree: tari_bulletproofs_plus::generators::pedersen_gens::ExtensionDegree,
) -> tari_bulletproofs_plus::PedersenGens<GpuRistretto> {
    let g = tari_bulletproofs_plus::ristretto::create_pedersen_gens_with_extension_degree(extension_degree);
    tari_bulletproofs_plus::PedersenGens {
        h_base: GpuRistretto(g.h_base),
        h_base_compressed: GpuCompressedRistretto(g.h_base_compressed),
        g_base_vec: g.g_base_vec.iter().map(|p| GpuRistretto(*p)).collect(),
        g_base_compressed_vec: g.g_base_compressed_vec.iter().map(|c| GpuCompressedRistretto(*c)).collect(),
        extension_degree,
    }
}

/// `RangeProof<GpuRistretto>`: the unpatched crate's prover and verifier with every multiscalar multiplication on the MI355X
pub type GpuRistrettoRangeProof = tari_bulletproofs_plus::range_proof::RangeProof<GpuRistretto>;

#[cfg(test)]
mod tests {
    //! differential: the device's MSMs against dalek's on the reference's own call shapes (`cargo test -- --ignored` on an MI355X)
    use super::*;
    use curve25519_dalek::ristretto::VartimeRistrettoPrecomputation;

    fn points(n: usize, tag: u8) -> Vec<RistrettoPoint> {
        (0..n).map(|i| {
            let mut b = [0u8; 64];
            b[0] = i as u8;
            b[1] = (i >> 8) as u8;
            b[63] = tag;
            RistrettoPoint::from_uniform_bytes(&b)
        }).collect()
    }

    #[test]
    #[ignore = "needs an MI355X"]
    fn mixed_msm_matches_dalek() {
        // the verifier's final check at configs[0]'s shape: 128 static generators, 18 dynamic points (src/range_proof.rs:1050-1057)
        let (stat, dynp) = (points(128, 1), points(18, 2));
        let ss: Vec<Scalar> = (0..128u64).map(|i| Scalar::from(i * i + 7)).collect();
        let ds: Vec<Scalar> = (0..18u64).map(|i| -Scalar::from(i + 3)).collect();
        let want = VartimeRistrettoPrecomputation::new(stat.iter()).vartime_mixed_multiscalar_mul(ss.iter(), ds.iter(), dynp.iter());
        let gs: Vec<GpuRistretto> = stat.iter().map(|p| GpuRistretto(*p)).collect();
        let gd: Vec<GpuRistretto> = dynp.iter().map(|p| GpuRistretto(*p)).collect();
        let got = GpuPrecomputation::new(gs.iter()).vartime_mixed_multiscalar_mul(ss.iter(), ds.iter(), gd.iter());
        assert_eq!(got.0.compress(), want.compress());
    }

    #[test]
    #[ignore = "needs an MI355X"]
    fn vartime_msm_matches_dalek() {
        // an L / R of the prover's first round at configs[4]'s shape: 260 terms (src/range_proof.rs:482-495)
        let pts = points(260, 3);
        let sc: Vec<Scalar> = (0..260u64).map(|i| Scalar::from(3 * i + 1) * Scalar::from(u64::MAX - i)).collect();
        let want = RistrettoPoint::vartime_multiscalar_mul(sc.iter(), pts.iter());
        let gp: Vec<GpuRistretto> = pts.iter().map(|p| GpuRistretto(*p)).collect();
        let got = GpuRistretto::vartime_multiscalar_mul(sc.iter(), gp.iter());
        assert_eq!(got.0.compress(), want.compress());
        assert_eq!(GpuRistretto::multiscalar_mul(sc.iter(), gp.iter()), got);
    }
}
