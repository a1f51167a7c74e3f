"""Sharding one reference batch across ranks (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

The path partitions by proof.  Two couplings exist in RangeProof::verify (src/range_proof.rs:756-1065):
  1. the batch weights come from ONE transcript over all proofs in order (:811,:849,:853,:894)
       -> all_gather of the 32 transcript-RNG bytes per proof, then every rank replays the (sequential) chain and
          keeps the weights of its own proofs;
  2. the final check is one group equation (:1050-1062)
       -> every rank reduces its proofs to ONE accumulator point; all_gather of the 128-byte accumulators
          (RCCL has no group-law reduction), rank-local sum in rank order, identity test.
`mode="shard"` instead treats each rank's proofs as an independent reference batch (what verify_batch's chunking
does): no data-path collective, only the verdicts are combined.
"""
import torch
import torch.distributed as dist

from . import api


class LocalEngineOps:
    """phase interface of one rank, backed by the C ABI (bpp_verify_phase1/2, bpp_accumulators_sum_is_identity)"""

    def __init__(self, resident_batch):
        self.rb = resident_batch

    def phase1(self):
        return self.rb.phase1()

    def phase2(self, weights32):
        return self.rb.phase2(weights32)

    def sum_is_identity(self, accumulators128):
        return api.accumulators_sum_is_identity(self.rb.engine, accumulators128)

    def verify_local(self):
        self.rb.verify(api.VerifyAction.VerifyOnly, chunk=0)
        return True


def _all_gather_bytes(data, device, group=None):
    """all_gather of equal-length byte strings; returns the concatenation in rank order"""
    world = dist.get_world_size(group)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    out = torch.empty(world * src.numel(), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, src, group=group) if device.type == "cuda" else \
        dist.all_gather(list(out.chunk(world)), src, group=group)
    return out.cpu().numpy().tobytes()


def verify_sharded(ops, n_local, device, mode="wide", group=None, weights_fn=api.weights_from_chain):
    """Verify the union of all ranks' resident batches.  Returns True (valid) or raises api.ProofError.

    ops: LocalEngineOps-like object for this rank's shard; every rank must hold the same number of proofs."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if mode == "shard":
        ok = 1
        err = None
        try:
            ops.verify_local()
        except api.ProofError as e:
            ok, err = 0, e
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if err is not None:
            raise err
        if int(flag.item()) == 0:
            raise api.ProofError(api.ProofErrorKind.VerificationFailed, "Range proof batch not valid (another shard)")
        return True
    # ---- wide: one reference batch over all ranks ----
    rng_local = ops.phase1()
    assert len(rng_local) == 32 * n_local
    rng_all = _all_gather_bytes(rng_local, device, group)
    weights_all = weights_fn(rng_all)  # sequential sponge: replayed by every rank, no broadcast needed
    acc = ops.phase2(weights_all[32 * n_local * rank:32 * n_local * (rank + 1)])
    accs = _all_gather_bytes(acc, device, group)
    assert len(accs) == 128 * world
    if not ops.sum_is_identity(accs):
        raise api.ProofError(api.ProofErrorKind.VerificationFailed, "Range proof batch not valid")
    return True
