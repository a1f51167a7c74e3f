"""Sharding one reference batch across ranks: a thin caller of the C ABI (include/bpp.h: bpp_comm_*, bpp_verify_sharded,
bpp_verify_sharded_wave).  One process per GPU; the collectives are RCCL calls on device buffers made by libbpp_hip.so.

The path partitions by proof.  Two couplings exist in RangeProof::verify (src/range_proof.rs:756-1065):
  1. the batch weights come from ONE transcript over all proofs in order (:811,:849,:853,:894)
       -> all_gather of the 32 transcript-RNG bytes per proof, then every rank replays the (sequential) chain and
          keeps the weights of its own proofs;
  2. the final check is one group equation (:1050-1062)
       -> every rank reduces its proofs to ONE accumulator point; all_gather of the 128-byte accumulators
          (RCCL has no group-law reduction), the same sum on every rank, identity test.
Errors: every rank reaches both collectives and every rank raises the same error, the one the single-process verify()
would have hit first, decided by NUMERIC tier (bpp.h BPP_TIER_*) then rank -- the rule lives in the C library
(csrc/upload_host.h: shard_local_trailer / shard_resolve) and is what `rehearse_sharded` below drives on CPU.

torch.distributed is used here for one thing only: handing the 128-byte ncclUniqueId of rank 0 to the other ranks
(any channel would do; a Rust caller uses its own).
"""
import ctypes
from ctypes import POINTER, byref, c_int, c_uint32, c_uint64, c_void_p

import torch
import torch.distributed as dist

from . import _lib, api

TRAILER = 128  # BPP_SHARD_TRAILER_BYTES


class ShardComm:
    """bpp_comm: one RCCL communicator + the staging of the two exchanges.  `engine` fixes the device."""

    def __init__(self, engine, rank, world, unique_id=None, local_group=None):
        self.engine, self.rank, self.world, self.lib = engine, rank, world, engine.lib
        self.handle = c_void_p()
        if local_group is not None:  # bpp_comm_create_local: ranks = threads of this process on one device (tests)
            rc = self.lib.bpp_comm_create_local(engine.ctx, int(local_group), rank, world, byref(self.handle))
        else:
            rc = self.lib.bpp_comm_create(engine.ctx, _lib_buf(unique_id), rank, world, byref(self.handle))
        api._check(rc, engine.ctx)

    @classmethod
    def from_callbacks(cls, engine, rank, world, all_gather):
        """bpp_comm_create_callbacks: `all_gather(send: bytes) -> bytes` (the ranks' blocks in rank order) is the transport; it is
        called on the thread that makes the verify call, in the same order on every rank"""
        self = cls.__new__(cls)
        self.engine, self.rank, self.world, self.lib = engine, rank, world, engine.lib
        self.handle = c_void_p()

        def trampoline(_user, send, recv, nbytes):
            try:
                got = all_gather(ctypes.string_at(send, nbytes))
                if len(got) != nbytes * world:
                    return 2
                ctypes.memmove(recv, got, len(got))
                return 0
            except Exception:  # noqa: BLE001 - an exception must not unwind through the C caller
                return 1
        self._cb = _lib.ALL_GATHER_FN(trampoline)  # kept alive as long as the communicator
        api._check(self.lib.bpp_comm_create_callbacks(engine.ctx, rank, world, ctypes.cast(self._cb, c_void_p), None, byref(self.handle)),
                   engine.ctx)
        return self

    @classmethod
    def from_process_group_gloo(cls, engine, group=None):
        """the caller-supplied transport over torch.distributed's CPU backend: what lets several ranks share ONE GPU (RCCL refuses
        two ranks on a device) and what a caller without RCCL uses.  COLLECTIVE when `group` is None: every communicator gets a
        process group of its OWN (dist.new_group over all ranks), so that several communicators driven from several threads never
        share one: their all_gathers have identical shapes, and on a shared group two of them issued in different orders on
        different ranks would pair up crosswise -- wrong bytes, or a hang -- without any error."""
        if group is None:
            group = dist.new_group(backend="gloo")
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def all_gather(send):
            src = torch.frombuffer(bytearray(send), dtype=torch.uint8)
            out = torch.empty(world * src.numel(), dtype=torch.uint8)
            dist.all_gather(list(out.chunk(world)), src, group=group)
            return out.numpy().tobytes()
        return cls.from_callbacks(engine, rank, world, all_gather)

    @staticmethod
    def unique_id():
        out = (ctypes.c_uint8 * 128)()
        rc = _lib.load().bpp_comm_unique_id(out)
        if rc != 0:
            raise api.EngineError("bpp_comm_unique_id failed (%d): RCCL not loadable" % rc)
        return bytes(out)

    @classmethod
    def from_process_group(cls, engine, group=None):
        """collective over `group`: rank 0's ncclUniqueId travels through torch.distributed, then ncclCommInitRank"""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(engine, rank, world, box[0])

    def _err(self):
        m = self.lib.bpp_comm_last_error(self.handle)
        return m.decode(errors="replace") if m else ""

    def verify(self, rb, counts):
        """bpp_verify_sharded: this rank's resident shard `rb` of ONE reference batch; counts[r] = proofs on rank r.
        Returns True or raises the ProofError every rank raises."""
        res = self.verify_wave([rb], counts)[0]
        return _raise(res)

    def verify_wave(self, rbs, counts):
        """bpp_verify_sharded_wave over k resident shards (each on its OWN engine / stream) -> list of result dicts"""
        k = len(rbs)
        ctxs = (c_void_p * k)(*[rb.engine.ctx for rb in rbs])
        hs = (c_uint64 * k)(*[rb.handle.value for rb in rbs])
        cn = (c_uint32 * self.world)(*counts)
        out = (_lib.ShardResult * k)()
        rc = self.lib.bpp_verify_sharded_wave(self.handle, ctxs, hs, k, cn, out)
        if rc != 0:
            raise api.EngineError("bpp_verify_sharded_wave failed (%d): %s" % (rc, self._err()))
        return [{"code": r.code, "tier": r.tier, "rank": r.rank, "index": r.index, "msg": r.msg.decode(errors="replace")} for r in out]

    def verify_groups(self, rb, n_groups, counts):
        """bpp_verify_sharded_groups: ONE resident batch holding this rank's shards of n_groups reference batches (group g =
        proofs [g c, (g+1) c), c = counts[rank]) -> list of n_groups result dicts"""
        cn = (c_uint32 * self.world)(*counts)
        out = (_lib.ShardResult * n_groups)()
        rc = self.lib.bpp_verify_sharded_groups(self.handle, rb.engine.ctx, rb.handle, n_groups, cn, out)
        if rc != 0:
            raise api.EngineError("bpp_verify_sharded_groups failed (%d): %s" % (rc, self._err()))
        return [{"code": r.code, "tier": r.tier, "rank": r.rank, "index": r.index, "msg": r.msg.decode(errors="replace")} for r in out]

    def verify_groups_wave(self, rbs, n_groups, counts):
        """bpp_verify_sharded_groups_wave: k grouped resident batches (each on its OWN engine) as a software pipeline of this
        one thread -> list (k) of lists (n_groups) of result dicts"""
        k = len(rbs)
        ctxs = (c_void_p * k)(*[rb.engine.ctx for rb in rbs])
        hs = (c_uint64 * k)(*[rb.handle.value for rb in rbs])
        cn = (c_uint32 * self.world)(*counts)
        out = (_lib.ShardResult * (k * n_groups))()
        rc = self.lib.bpp_verify_sharded_groups_wave(self.handle, ctxs, hs, k, n_groups, cn, out)
        if rc != 0:
            raise api.EngineError("bpp_verify_sharded_groups_wave failed (%d): %s" % (rc, self._err()))
        res = [{"code": r.code, "tier": r.tier, "rank": r.rank, "index": r.index, "msg": r.msg.decode(errors="replace")} for r in out]
        return [res[i * n_groups:(i + 1) * n_groups] for i in range(k)]

    def set_timeout(self, ms):
        """deadline of every wait for a collective (bpp_comm_set_timeout): past it the call returns BPP_ERR_COMM"""
        api._check(self.lib.bpp_comm_set_timeout(self.handle, int(ms)), None)

    def last_timing(self):
        """host wall-clock split of the last wave (ms)"""
        t = _lib.ShardTiming()
        self.lib.bpp_comm_last_timing(self.handle, byref(t))
        return {n: getattr(t, n) for n, _ in _lib.ShardTiming._fields_}

    def close(self):
        if self.handle:
            self.lib.bpp_comm_destroy(self.handle)
            self.handle = c_void_p()


def _lib_buf(data):
    return (ctypes.c_uint8 * len(data)).from_buffer_copy(data)


def _raise(res):
    if res["code"] == 0:
        return True
    if res["code"] > 0:
        e = api.ProofError(res["code"], res["msg"])
        e.tier, e.rank, e.index = res["tier"], res["rank"], res["index"]
        raise e
    raise api.EngineError("bpp engine error %d: %s" % (res["code"], res["msg"]))


def verify_shard_mode(rb, device, group=None):
    """each rank's proofs as an independent reference batch (what verify_batch's chunking does): no
    data-path collective, only the verdicts are combined"""
    ok, err = 1, None
    try:
        rb.verify(api.VerifyAction.VerifyOnly, chunk=0)
    except api.ProofError as e:
        ok, err = 0, e
    # (the verdicts travel over whatever backend the caller's group has: a CPU tensor unless that is NCCL / RCCL)
    flag = torch.tensor([ok], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if err is not None:
        raise err
    if int(flag.item()) == 0:
        raise api.ProofError(api.ProofErrorKind.VerificationFailed, "Range proof batch not valid (another shard)")
    return True


# ---------------------------------------------------------------------------------------------------------------------
# CPU rehearsal of the same protocol (tests/test_dist_gloo.py, world size 2 over gloo): the per-rank kernels need a GPU, so
# a stand-in supplies their OUTPUTS (per-proof status words, RNG bytes, accumulator); everything that decides -- which
# finding a rank reports, which one wins, the weight chain -- is the product's C code, and the transport is
# torch.distributed instead of RCCL.  Not a product path: bpp_verify_sharded is.

def local_trailer(defer, status, rounds_bad, first_index):
    """bpp_shard_local_trailer: the engine's own order of checks over a rank's per-proof facts -> 128-byte finding"""
    n = len(status)
    out = (ctypes.c_uint8 * TRAILER)()
    d = (ctypes.c_uint8 * max(n, 1))(*defer) if defer is not None else None
    st = (c_uint32 * max(n, 1))(*status)
    rb = (ctypes.c_uint8 * max(n, 1))(*rounds_bad)
    rc = _lib.load().bpp_shard_local_trailer(d, st, rb, n, first_index, out)
    assert rc == 0
    return bytes(out)


def fault_trailer(code, first_index, msg):
    out = (ctypes.c_uint8 * TRAILER)()
    assert _lib.load().bpp_shard_trailer(255, code, first_index, msg.encode()[:110], out) == 0
    return bytes(out)


def resolve(trailers):
    """bpp_shard_resolve over the ranks' trailers -> result dict (code 0 = all clean)"""
    world = len(trailers)
    tier, rank, index = c_int(), c_int(), c_uint32()
    err = ctypes.create_string_buffer(160)
    code = _lib.load().bpp_shard_resolve(_lib_buf(b"".join(trailers)), TRAILER, world, byref(tier), byref(rank), byref(index), err, 160)
    return {"code": code, "tier": tier.value, "rank": rank.value, "index": index.value, "msg": err.value.decode(errors="replace")}


def _all_gather_bytes(data, device, group=None):
    """all_gather of equal-length byte strings; returns the concatenation in rank order"""
    world = dist.get_world_size(group)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    out = torch.empty(world * src.numel(), dtype=torch.uint8, device=device)
    dist.all_gather(list(out.chunk(world)), src, group=group)
    return out.cpu().numpy().tobytes()


def rehearse_sharded(ops, n_local, device, group=None, weights_fn=api.weights_from_chain):
    """the two exchanges of bpp_verify_sharded with `ops` standing in for the kernels:
         ops.phase1_facts() -> (rng bytes, defer | None, status words, rounds_bad)     ops.phase2(weights32) -> 128-byte accumulator
         ops.sum_is_identity(accumulators128) -> bool
    As in the library: only the RNG bytes cross before the weights exist; what a rank found travels with its accumulator,
    and a finding of any rank (lowest tier, then lowest rank) comes before the final check.  Whatever a rank's stand-in
    raises, the rank still reaches both collectives (an engine-fault finding travels instead)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first = n_local * rank
    try:
        rng_local, defer, status, rounds_bad = ops.phase1_facts()
        assert len(rng_local) == 32 * n_local
        tr = local_trailer(defer, status, rounds_bad, first)
    except Exception as e:  # noqa: BLE001 - anything at all: the collective below must still be entered
        rng_local, tr = bytes(32 * n_local), fault_trailer(-1, first, "%s: %s" % (type(e).__name__, e))
    gathered = _all_gather_bytes(rng_local, device, group)
    weights_all = weights_fn(gathered)  # sequential sponge over ALL proofs: replayed by every rank, no broadcast needed
    acc = bytes(128)
    if tr == bytes(TRAILER):  # nothing found here: this rank's share of the final sum (the library runs it regardless;
        try:                  # the stand-in cannot compute on proofs it has already rejected)
            acc = ops.phase2(weights_all[32 * n_local * rank:32 * n_local * (rank + 1)])
            assert len(acc) == 128
        except Exception as e:  # noqa: BLE001
            acc, tr = bytes(128), fault_trailer(-1, first, "%s: %s" % (type(e).__name__, e))
    both = _all_gather_bytes(acc + tr, device, group)
    stride = 128 + TRAILER
    res = resolve([both[stride * r + 128:stride * (r + 1)] for r in range(world)])
    if res["code"] != 0:
        return _raise(res)
    if not ops.sum_is_identity(b"".join(both[stride * r:stride * r + 128] for r in range(world))):
        e = api.ProofError(api.ProofErrorKind.VerificationFailed, "Range proof batch not valid")
        e.tier, e.rank, e.index = 7, -1, 0
        raise e
    return True
