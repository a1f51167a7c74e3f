"""CPU suite: the N>1 path (bulletproofs-plus_amd/dist.py) with world_size 2 over gloo.

The collectives, shard slicing and weight-chain replay are the product's; the per-rank phase1/phase2 kernels need a GPU,
so here they are stood in by the oracle (tests may use it as the checker).  The weight chain itself is the product's
host function bpp_weights_from_chain."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.pyref import curve as C
from oracle.pyref import merlin as M
from oracle.pyref import protocol as O
from tests.helpers import make_oracle_batch, sb


def _pt_bytes(p):
    return b"".join((v % C.P).to_bytes(32, "little") for v in (p.X, p.Y, p.Z, p.T))


def _pt_from(b):
    return C.Point(*[int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(4)])


def _with_member_a(proof, enc):
    """the proof with its member A replaced by the 32-byte encoding `enc` (wire offset 1 + 32 t)"""
    raw = bytearray(proof.to_bytes())
    raw[1 + 32 * proof.extension_degree:1 + 32 * proof.extension_degree + 32] = enc
    return O.RangeProof.from_bytes(bytes(raw))


class OracleOps:
    def __init__(self, case, lo, hi, tamper=None):
        self.sts = case.o_statements_public[lo:hi]
        self.proofs = case.o_proofs[lo:hi]
        if tamper == "promise":  # only the final MSM notices
            s = self.sts[0]
            self.sts = [O.RangeStatement(s.generators, s.commitments, [(v + 1 if v is not None else 1) for v in s.minimum_value_promises], None)] + self.sts[1:]
        elif tamper == "identity":  # PASS-1 error: identity element appended to the transcript -> VerificationFailed
            self.proofs = [self.proofs[0], _with_member_a(self.proofs[1], bytes(32))]
        elif tamper == "badpoint":  # PASS-2 error: A does not decode -> InvalidArgument
            self.proofs = [_with_member_a(self.proofs[0], b"\x01" + bytes(31)), self.proofs[1]]
        self.label = case.label
        self.weights_used = None

    def _tr(self):
        return [M.Transcript(self.label) for _ in self.proofs]

    def phase1(self):
        """like bpp_verify_phase1: the rng bytes, or the first error of this shard in the reference's order of checks"""
        api = importlib.import_module("bulletproofs-plus_amd")
        tr = {}
        try:
            O.verify(self._tr(), self.sts, self.proofs, O.VERIFY_ONLY, trace=tr, check=False)
        except O.ProofError as e:
            raise api.ProofError(e.kind, e.msg)
        return b"".join(tr["rng_outputs"])

    def phase2(self, weights32):
        w = [int.from_bytes(weights32[32 * i:32 * i + 32], "little") for i in range(len(self.proofs))]
        self.weights_used = w
        tr = {}
        O.verify(self._tr(), self.sts, self.proofs, O.VERIFY_ONLY, trace=tr, weights_override=w, check=False)
        return _pt_bytes(tr["accumulator"])

    def sum_is_identity(self, accs):
        acc = C.Point.identity()
        for i in range(len(accs) // 128):
            acc = acc + _pt_from(accs[128 * i:128 * i + 128])
        return acc == C.Point.identity()

    def verify_local(self):
        api = importlib.import_module("bulletproofs-plus_amd")
        try:
            O.verify(self._tr(), self.sts, self.proofs, O.VERIFY_ONLY)
        except O.ProofError as e:  # the engine raises the product's ProofError
            raise api.ProofError(e.kind, e.msg)
        return True


# (mode, what rank 0 holds, what rank 1 holds, ProofError kind every rank must see | None)
SCENARIOS = [("wide", None, None, None), ("wide", None, "promise", 1), ("shard", None, None, None),
             ("shard", None, "promise", 1),
             # a phase-1 failure on ONE rank: every rank still reaches both collectives and raises the same error
             ("wide", None, "identity", 1), ("wide", None, "badpoint", 2), ("wide", "badpoint", None, 2),
             # PASS-1 errors of ANY proof come before PASS-2 errors (src/range_proof.rs:816-850 vs :859-888)
             ("wide", "badpoint", "identity", 1),
             # and the ranks are still in step afterwards
             ("wide", None, None, None)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bpp = importlib.import_module("bulletproofs-plus_amd")
        dmod = importlib.import_module("bulletproofs-plus_amd.dist")
        case = make_oracle_batch(8, [1, 1, 1, 1], 1, seed=b"gloo")
        tr = {}
        O.verify([M.Transcript(case.label) for _ in case.o_proofs], case.o_statements_public, case.o_proofs,
                 O.VERIFY_ONLY, trace=tr)
        out = []
        for mode, t0, t1, _want in SCENARIOS:
            ops = OracleOps(case, 2 * rank, 2 * rank + 2, tamper=(t0, t1)[rank])
            try:
                res = ("ok", dmod.verify_sharded(ops, 2, torch.device("cpu"), mode=mode))
            except bpp.ProofError as e:
                res = ("err", int(e.kind))
            if mode == "wide" and t0 is None and t1 is None:
                # the weights each rank used are exactly the single-process reference weights of its proofs
                assert ops.weights_used == tr["weights"][2 * rank:2 * rank + 2]
            out.append(res)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i, (mode, t0, t1, want) in enumerate(SCENARIOS):
        for rank in (0, 1):
            if want is not None:
                assert res[rank][i] == ("err", want), (mode, t0, t1, rank, res[rank][i])  # the same error on every rank
            else:
                assert res[rank][i] == ("ok", True), (mode, t0, t1, rank, res[rank][i])
