//! Safe wrappers over libbpp_hip.so for the two calls the reference crate routes through the GPU:
//! `RangeProof::verify_batch` (src/range_proof.rs:712-752) and `RangeProof::prove_with_rng` (:232-608).
//!
//! Everything crosses as bytes (32-byte ristretto255 encodings, 32-byte canonical scalars, `RangeProof::to_bytes()`), so
//! this crate does not depend on curve25519-dalek or on the reference crate; `patch/range_proof_gpu.rs` is the code that
//! lives INSIDE the reference crate (its fields and `RangeProofTranscript` are private) and calls into here.
pub mod ffi;

use core::ffi::c_int;
use std::{ffi::CStr, ptr, sync::Mutex};

/// Mirror of `tari_bulletproofs_plus::errors::ProofError` (src/errors.rs:11-28) plus engine faults.
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum GpuError {
    VerificationFailed(String),
    InvalidArgument(String),
    InvalidLength(String),
    InvalidBlake2b,
    SizeOverflow,
    /// HIP failure, no gfx950 device, bad handle: no reference analogue
    Engine(i32, String),
}

fn map_rc(rc: c_int, msg: String) -> Result<(), GpuError> {
    match rc {
        ffi::BPP_OK => Ok(()),
        ffi::BPP_ERR_VERIFICATION_FAILED => Err(GpuError::VerificationFailed(msg)),
        ffi::BPP_ERR_INVALID_ARGUMENT => Err(GpuError::InvalidArgument(msg)),
        ffi::BPP_ERR_INVALID_LENGTH => Err(GpuError::InvalidLength(msg)),
        ffi::BPP_ERR_INVALID_BLAKE2B => Err(GpuError::InvalidBlake2b),
        ffi::BPP_ERR_SIZE_OVERFLOW => Err(GpuError::SizeOverflow),
        other => Err(GpuError::Engine(other, msg)),
    }
}

/// VerifyAction (src/range_proof.rs:46-54)
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
#[repr(i32)]
pub enum Action {
    VerifyOnly = 0,
    RecoverAndVerify = 1,
    RecoverOnly = 2,
}

/// One device context (device + stream + work buffers).  Calls on one context are serialised by the engine; use one
/// context per host thread for concurrency.  `Params` handles are shared between contexts (Arc semantics).
pub struct Engine {
    ctx: *mut ffi::bpp_ctx,
}
unsafe impl Send for Engine {}

impl Engine {
    pub fn new(device_id: i32) -> Result<Self, GpuError> {
        let mut ctx = ptr::null_mut();
        map_rc(unsafe { ffi::bpp_ctx_create(&mut ctx, device_id) }, "bpp_ctx_create: a gfx950 device is required".into())?;
        Ok(Engine { ctx })
    }

    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(ffi::bpp_ctx_last_error(self.ctx)).to_string_lossy().into_owned() }
    }

    /// RangeParameters::init (src/range_parameters.rs:32-58); `h_base` / `g_bases` = None: the reference's defaults
    pub fn params(&self, bit_length: usize, max_aggregation: usize, extension_degree: usize, h_base: Option<&[u8; 32]>,
                  g_bases: Option<&[u8]>) -> Result<Params, GpuError> {
        let mut handle = 0u64;
        let rc = unsafe {
            ffi::bpp_params_create(self.ctx, bit_length as u32, max_aggregation as u32, extension_degree as u32,
                                   h_base.map_or(ptr::null(), |h| h.as_ptr()), g_bases.map_or(ptr::null(), |g| g.as_ptr()), &mut handle)
        };
        map_rc(rc, self.last_error())?;
        Ok(Params { handle, extension_degree })
    }

    /// `RangeProof::verify_batch`: every `chunk` consecutive items are one reference batch (256 = MAX_RANGE_PROOF_BATCH_SIZE,
    /// 0 = the whole input).  Returns per item `Some(mask blindings, t x 32 bytes)` or `None`.
    pub fn verify_batch(&self, params: &Params, items: &[VerifyItem<'_>], action: Action, chunk: usize)
                        -> Result<Vec<Option<Vec<[u8; 32]>>>, GpuError> {
        let t = params.extension_degree;
        let present_in: Vec<Vec<u8>> = items.iter().map(|i| i.min_values.iter().map(|v| v.is_some() as u8).collect()).collect();
        let mins: Vec<Vec<u64>> = items.iter().map(|i| i.min_values.iter().map(|v| v.unwrap_or(0)).collect()).collect();
        let raw: Vec<ffi::bpp_verify_item> = items.iter().enumerate().map(|(k, i)| ffi::bpp_verify_item {
            proof: i.proof.as_ptr(),
            proof_len: i.proof.len(),
            commitments32: i.commitments.as_ptr(),
            m: (i.commitments.len() / 32) as u32,
            min_values: mins[k].as_ptr(),
            min_present: present_in[k].as_ptr(),
            seed_nonce32: i.seed_nonce.map_or(ptr::null(), |s| s.as_ptr()),
            transcript_state: i.transcript_state.map_or(ptr::null(), |s| s.as_ptr()),
            transcript_label: i.transcript_label.as_ptr(),
            label_len: i.transcript_label.len(),
        }).collect();
        let mut masks = vec![0u8; items.len() * t * 32];
        let mut present = vec![0u8; items.len()];
        let mut err = [0 as core::ffi::c_char; 256];
        let rc = unsafe {
            ffi::bpp_verify_batch(self.ctx, params.handle, raw.as_ptr(), raw.len(), action as c_int, chunk, masks.as_mut_ptr(),
                                  present.as_mut_ptr(), err.as_mut_ptr(), err.len())
        };
        map_rc(rc, unsafe { CStr::from_ptr(err.as_ptr()) }.to_string_lossy().into_owned())?;
        Ok((0..items.len()).map(|k| if present[k] != 0 {
            Some((0..t).map(|j| { let mut b = [0u8; 32]; b.copy_from_slice(&masks[(k * t + j) * 32..(k * t + j + 1) * 32]); b }).collect())
        } else { None }).collect())
    }

    /// n x `RangeProof::prove_with_rng` in one call; all items share the aggregation factor.  Returns `to_bytes()` of each proof.
    pub fn prove_batch(&self, params: &Params, items: &[ProveItem<'_>]) -> Result<Vec<Vec<u8>>, GpuError> {
        let present_in: Vec<Vec<u8>> = items.iter().map(|i| i.min_values.iter().map(|v| v.is_some() as u8).collect()).collect();
        let mins: Vec<Vec<u64>> = items.iter().map(|i| i.min_values.iter().map(|v| v.unwrap_or(0)).collect()).collect();
        let raw: Vec<ffi::bpp_prove_item> = items.iter().enumerate().map(|(k, i)| ffi::bpp_prove_item {
            values: i.values.as_ptr(),
            blindings32: i.blindings.as_ptr(),
            commitments32: i.commitments.as_ptr(),
            m: i.values.len() as u32,
            min_values: mins[k].as_ptr(),
            min_present: present_in[k].as_ptr(),
            seed_nonce32: i.seed_nonce.map_or(ptr::null(), |s| s.as_ptr()),
            transcript_state: i.transcript_state.map_or(ptr::null(), |s| s.as_ptr()),
            transcript_label: i.transcript_label.as_ptr(),
            label_len: i.transcript_label.len(),
            rng_bytes: i.rng_bytes.as_ptr(),
            rng_len: i.rng_bytes.len(),
        }).collect();
        let stride = 1 + 32 * (6 + 5 + 2 * 12);
        let mut out = vec![0u8; stride * items.len()];
        let mut plen = 0usize;
        let mut err = [0 as core::ffi::c_char; 256];
        let rc = unsafe {
            ffi::bpp_prove_batch(self.ctx, params.handle, raw.as_ptr(), raw.len(), out.as_mut_ptr(), stride, &mut plen, err.as_mut_ptr(), err.len())
        };
        map_rc(rc, unsafe { CStr::from_ptr(err.as_ptr()) }.to_string_lossy().into_owned())?;
        Ok((0..items.len()).map(|k| out[k * stride..k * stride + plen].to_vec()).collect())
    }

    /// Arc::clone of a parameter set created on another context of the same device
    pub fn retain(&self, params: &Params) -> Result<Params, GpuError> {
        map_rc(unsafe { ffi::bpp_params_retain(self.ctx, params.handle) }, self.last_error())?;
        Ok(Params { handle: params.handle, extension_degree: params.extension_degree })
    }

    pub fn release(&self, params: Params) {
        unsafe { ffi::bpp_params_destroy(self.ctx, params.handle) };
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { ffi::bpp_ctx_destroy(self.ctx) }
    }
}

/// Device-resident generator tables of one `RangeParameters` (src/range_parameters.rs:20-30): process-wide, shared.
pub struct Params {
    handle: u64,
    extension_degree: usize,
}

pub struct VerifyItem<'a> {
    pub proof: &'a [u8],
    /// m x 32 bytes: statement.commitments_compressed
    pub commitments: &'a [u8],
    pub min_values: &'a [Option<u64>],
    pub seed_nonce: Option<&'a [u8; 32]>,
    /// a fresh `Transcript::new(label)` ...
    pub transcript_label: &'a [u8],
    /// ... or the 203-byte STROBE state of an arbitrary transcript
    pub transcript_state: Option<&'a [u8; 203]>,
}

pub struct ProveItem<'a> {
    pub values: &'a [u64],
    /// m x t x 32 bytes
    pub blindings: &'a [u8],
    pub commitments: &'a [u8],
    pub min_values: &'a [Option<u64>],
    pub seed_nonce: Option<&'a [u8; 32]>,
    pub transcript_label: &'a [u8],
    pub transcript_state: Option<&'a [u8; 203]>,
    /// (rounds + 3) x 32 bytes drawn from the caller's `CryptoRngCore`, in order (what `TranscriptRngBuilder::finalize` pulls)
    pub rng_bytes: &'a [u8],
}

/// Process-wide default engine on device 0 for the in-crate patch (one context; callers that want concurrency hold their own).
pub fn default_engine() -> &'static Mutex<Engine> {
    use std::sync::OnceLock;
    static ENGINE: OnceLock<Mutex<Engine>> = OnceLock::new();
    ENGINE.get_or_init(|| Mutex::new(Engine::new(0).expect("bpp-gpu-shim: no usable gfx950 device")))
}
