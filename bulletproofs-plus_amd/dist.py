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
    # Every rank enters BOTH collectives whatever happens locally: a rank whose phase raised sends a placeholder plus a
    # 32-byte status trailer, and all ranks raise the same error afterwards (a rank that raised before the all_gather
    # would leave the others blocked in it, or -- if the caller carried on -- pair collectives of different batches).
    err = None
    try:
        rng_local = ops.phase1()
        assert len(rng_local) == 32 * n_local
    except api.ProofError as e:
        err, rng_local = e, bytes(32 * n_local)
    gathered = _all_gather_bytes(rng_local + _status_trailer(err), device, group)
    stride = 32 * n_local + 32
    _raise_first([gathered[stride * r + 32 * n_local:stride * (r + 1)] for r in range(world)])
    rng_all = b"".join(gathered[stride * r:stride * r + 32 * n_local] for r in range(world))
    weights_all = weights_fn(rng_all)  # sequential sponge: replayed by every rank, no broadcast needed
    try:
        acc = ops.phase2(weights_all[32 * n_local * rank:32 * n_local * (rank + 1)])
        assert len(acc) == 128
    except api.ProofError as e:
        err, acc = e, bytes(128)
    accs = _all_gather_bytes(acc + _status_trailer(err), device, group)
    assert len(accs) == 160 * world
    _raise_first([accs[160 * r + 128:160 * (r + 1)] for r in range(world)])
    if not ops.sum_is_identity(b"".join(accs[160 * r:160 * r + 128] for r in range(world))):
        raise api.ProofError(api.ProofErrorKind.VerificationFailed, "Range proof batch not valid")
    return True


def _tier(err):
    """position of the failing check inside RangeProof::verify (src/range_proof.rs:756-1065): the consistency loops
    (:637-659 degree, :674-682 promises), statement points, PASS 1 over ALL proofs (:816-850), then PASS 2 in proof order"""
    if "Inconsistent extension degree" in err.msg:
        return 0
    if "Minimum value promise" in err.msg:
        return 1
    if "Statement commitment" in err.msg:
        return 2
    if err.kind == api.ProofErrorKind.VerificationFailed:
        return 3
    return 4


def _status_trailer(err):
    """32 bytes: [0] = 0 ok / 1 + tier, [1] = ProofError kind, [2] = message length, [3..] message"""
    if err is None:
        return bytes(32)
    msg = err.msg.encode()[:29]
    return (bytes([1 + _tier(err), int(err.kind), len(msg)]) + msg).ljust(32, b"\0")


def _raise_first(trailers):
    """the error the single-process verify() would have hit first: lowest tier, then lowest rank (shards are contiguous
    in proof order); identical on every rank"""
    failed = [(t[0], r) for r, t in enumerate(trailers) if t[0]]
    if not failed:
        return
    _, r = min(failed)
    t = trailers[r]
    raise api.ProofError(t[1], t[3:3 + t[2]].decode(errors="replace") + " (rank %d)" % r)
