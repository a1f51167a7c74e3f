"""CPU suite: the N>1 protocol of bpp_verify_sharded with world_size 2 over gloo (dist.rehearse_sharded).

What decides is the product's C code, loaded from libbpp_hip.so: which finding a rank reports from its per-proof status
words (bpp_shard_local_trailer: the engine's own order of checks, numeric tiers, the engine's own messages), which finding
wins across ranks (bpp_shard_resolve) and the weight chain (bpp_weights_from_chain).  The per-rank kernels need a GPU, so
their OUTPUTS (status words, transcript-RNG bytes, accumulators) are stood in by the oracle; the transport is gloo where
the product uses RCCL.  Every scenario's outcome is compared with what the single-process oracle verify() raises on the
union of the shards."""
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
        self.fault = tamper == "fault"

    def _tr(self):
        return [M.Transcript(self.label) for _ in self.proofs]

    def phase1_facts(self):
        """what bpp_verify_sharded's phase 1 leaves on a rank: the RNG bytes and, per proof, the kernels' status word
        (layout.h: 1 = PASS-1 failure, 2 = a proof point does not decode), the deferred consistency bits and the L/R
        finding recorded at upload.  Classified proof by proof from the oracle (test infrastructure may read its messages)."""
        rng, status, rounds_bad, defer = [], [], [], []
        for st, pr in zip(self.sts, self.proofs):
            tr, word, rb = {}, 0, 0
            try:
                O.verify([M.Transcript(self.label)], [st], [pr], O.VERIFY_ONLY, trace=tr, check=False)
            except O.ProofError as e:
                if e.kind == O.VERIFICATION_FAILED:
                    word |= 1
                elif "canonical encoding of a point" in e.msg:
                    word |= 2
                elif e.kind == O.INVALID_LENGTH:
                    rb = 3
                elif e.kind == O.SIZE_OVERFLOW:
                    rb = 5
                else:
                    raise
            rng.append(tr["rng_outputs"][0] if tr.get("rng_outputs") else bytes(32))
            status.append(word)
            rounds_bad.append(rb)
            defer.append(0)
        if self.fault:
            raise RuntimeError("simulated engine fault")
        return b"".join(rng), defer, status, rounds_bad

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


# (what rank 0 holds, what rank 1 holds, expected (kind, tier, rank, index in the whole batch) on EVERY rank | None)
SCENARIOS = [(None, None, None),
             (None, "promise", (1, 7, -1, 0)),            # only the final check notices: tier MSM
             # a phase-1 finding on ONE rank: every rank still reaches both collectives and raises the same error
             (None, "identity", (1, 5, 1, 3)),            # PASS 1, proof 1 of rank 1 = proof 3 of the batch
             (None, "badpoint", (2, 6, 1, 2)), ("badpoint", None, (2, 6, 0, 0)),
             # PASS-1 findings of ANY proof come before PASS-2 findings (src/range_proof.rs:816-850 vs :859-888)
             ("badpoint", "identity", (1, 5, 1, 3)),
             # same tier on both ranks: the lower rank's (earlier proofs)
             ("badpoint", "badpoint", (2, 6, 0, 0)),
             # a rank whose engine faults still reaches the collective; a real finding elsewhere wins over it
             ("fault", None, "engine"), ("fault", "identity", (1, 5, 1, 3)),
             # and the ranks are still in step afterwards
             (None, None, None)]


def _single_process_kind(case, t0, t1):
    """what verify() raises on the union of the two shards (the reference's own order of checks)"""
    a, b = OracleOps(case, 0, 2, tamper=t0), OracleOps(case, 2, 4, tamper=t1)
    try:
        O.verify([M.Transcript(case.label) for _ in range(4)], a.sts + b.sts, a.proofs + b.proofs, O.VERIFY_ONLY)
    except O.ProofError as e:
        return int(e.kind)
    return None


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
        for t0, t1, _want in SCENARIOS:
            ops = OracleOps(case, 2 * rank, 2 * rank + 2, tamper=(t0, t1)[rank])
            try:
                res = ("ok", dmod.rehearse_sharded(ops, 2, torch.device("cpu")))
            except bpp.ProofError as e:
                res = ("err", int(e.kind), e.tier, e.rank, e.index, e.msg)
            except bpp.EngineError as e:
                res = ("engine", str(e))
            if t0 is None and t1 is None:
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
    case = make_oracle_batch(8, [1, 1, 1, 1], 1, seed=b"gloo")
    for i, (t0, t1, want) in enumerate(SCENARIOS):
        assert res[0][i] == res[1][i], (t0, t1, res[0][i], res[1][i])  # the same outcome on every rank
        got = res[0][i]
        if want is None:
            assert got == ("ok", True), (t0, t1, got)
        elif want == "engine":
            assert got[0] == "engine" and "rank 0" in got[1], got
        else:
            assert got[:5] == ("err",) + want, (t0, t1, got)
            # ... and it is the kind the single-process verify() raises on the union of the shards
            if "fault" not in (t0, t1):
                assert _single_process_kind(case, t0, t1) == want[0]
    # the messages are the engine's own (csrc/upload_host.h), carried by the trailer
    msgs = {s[:2]: res[0][i][5] for i, s in enumerate(SCENARIOS) if res[0][i][0] == "err"}
    assert msgs[(None, "identity")].startswith("Identity element cannot be added to the transcript")
    assert msgs[(None, "badpoint")].startswith("A proof member was not the canonical encoding of a point")
    assert msgs[(None, "badpoint")].endswith("(rank 1)")


def test_tier_rule_of_the_library():
    """bpp_shard_local_trailer / bpp_shard_resolve directly: every tier against every other across two ranks, in the
    reference's order of checks; ties go to the lower rank; an engine fault only wins when nothing else was found"""
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    facts = {  # name -> (defer, status, rounds_bad) of a 3-proof shard, expected (tier, code, index in the shard)
        "clean": ([0, 0, 0], [0, 0, 0], [0, 0, 0], None),
        "degree": ([0, 0, 1], [1, 2, 0], [0, 3, 0], (2, 2, 2)),        # deferred degree beats everything the kernels saw
        "promise": ([2, 0, 0], [0, 0, 2], [0, 0, 0], (3, 3, 0)),
        "degree+promise": ([2, 1, 0], [0, 0, 0], [0, 0, 0], (2, 2, 1)),  # every degree before any promise (:637-682)
        "commit": ([0, 0, 0], [1, 4, 2], [0, 0, 0], (4, 2, 1)),
        "pass1": ([0, 0, 0], [2, 0, 1], [3, 0, 0], (5, 1, 2)),           # PASS 1 of proof 2 before PASS 2 of proof 0
        "decode": ([0, 0, 0], [0, 2, 2], [0, 0, 5], (6, 2, 1)),
        "length": ([0, 0, 0], [0, 0, 2], [0, 3, 0], (6, 3, 1)),          # PASS 2 is in proof order: L/R count of proof 1
        "decode-before-length": ([0, 0, 0], [0, 2, 0], [0, 3, 0], (6, 2, 1)),  # same proof: decompression first (:859-888)
        "overflow": ([0, 0, 0], [0, 0, 0], [0, 0, 5], (6, 5, 2)),
    }
    trailers = {}
    for name, (defer, status, rb, want) in facts.items():
        for first in (0, 3):
            t = dmod.local_trailer(defer, status, rb, first)
            r = dmod.resolve([t])
            if want is None:
                assert r["code"] == 0 and r["tier"] == 0 and t == bytes(128)
            else:
                assert (r["tier"], r["code"], r["index"], r["rank"]) == (want[0], want[1], first + want[2], 0), (name, r)
            trailers[(name, first)] = t
    trailers[("fault", 0)] = dmod.fault_trailer(-1, 0, "hipErrorLaunchFailure")
    trailers[("fault", 3)] = dmod.fault_trailer(-1, 3, "hipErrorLaunchFailure")
    tier_of = {n: (f[3][0] if f[3] else 0) for n, f in facts.items()}
    tier_of["fault"] = 255
    for a in tier_of:
        for b in tier_of:
            r = dmod.resolve([trailers[(a, 0)], trailers[(b, 3)]])
            ta, tb = tier_of[a], tier_of[b]
            if ta == 0 and tb == 0:
                assert r["code"] == 0
                continue
            win = 0 if (ta and (not tb or ta <= tb)) else 1
            assert r["rank"] == win and r["tier"] == (ta, tb)[win], (a, b, r)
            assert r["msg"].endswith("(rank %d)" % win)
