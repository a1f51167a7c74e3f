"""bulletproofs-plus_amd: MI355X (gfx950) engine for the Bulletproofs+ range-proof hot path.

The directory name carries a hyphen (it mirrors the reference repository's name), so import it with
    bpp = importlib.import_module("bulletproofs-plus_amd")
The compute path is libbpp_hip.so (hand-written HIP, C ABI in include/bpp.h); this package is the loader plus the
host-side mirror of the reference's RangeProof / RangeStatement / RangeParameters interface.
"""
from . import _build, _lib  # noqa: F401
from .api import (  # noqa: F401
    CommitmentOpening, EngineError, Engine, ExtendedMask, ExtensionDegree, MAX_RANGE_PROOF_BATCH_SIZE, PedersenGens, Precomputation,
    ProofError, ProofErrorKind, RangeParameters, RangeProof, RangeStatement, RangeWitness, ResidentBatch, Transcript, VerifyAction,
    accumulators_sum_is_identity, create_pedersen_gens_with_extension_degree, host_pool_cpu_ns, host_threads, shader_clock_ghz, verify_batch_with_challenges, weights_from_chain, weights_from_chains,
)

build = _build.build
