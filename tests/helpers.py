"""Test-side glue: builds the same batch for the product (bytes through the C ABI) and for the oracle (big ints).

Inputs follow the reference's own test recipe (tests/ristretto.rs:152-227, benches/range_proof.rs:206-262): value =
next_u64 % 2^(n-1), one random non-zero blinding repeated t times, seed nonce iff m == 1, transcript label
"BatchedRangeProofTest".  The PRNG is SHAKE256-based (the reference's ChaCha12 stream is not reproducible here and no
test of the reference depends on its actual bytes)."""
import hashlib

from oracle.pyref import curve as C
from oracle.pyref import merlin as M
from oracle.pyref import protocol as O

LABEL = b"BatchedRangeProofTest"


class Prng:
    def __init__(self, seed):
        self._s = hashlib.shake_256(seed)
        self._off = 0

    def fill_bytes(self, n):
        out = self._s.digest(self._off + n)[self._off:]
        self._off += n
        return out

    def next_u64(self):
        return int.from_bytes(self.fill_bytes(8), "little")


def sb(x):
    return C.scalar_bytes(x)


class Case:
    pass


def make_oracle_batch(bit_length, aggregation, extension_degree, seed=b"8675309", strategy="third", m_max=None,
                      label=LABEL):
    """Oracle-side batch: parameters, statements, witnesses and proofs made by the oracle prover."""
    rng = Prng(seed)
    c = Case()
    c.bit_length, c.aggregation, c.t = bit_length, list(aggregation), extension_degree
    c.label = label
    c.m_max = m_max or max(aggregation)
    c.o_params = O.RangeParameters(bit_length, c.m_max, O.PedersenGens(extension_degree))
    c.o_statements_private, c.o_statements_public, c.o_proofs, c.o_witnesses, c.expected_masks = [], [], [], [], []
    for m in aggregation:
        openings, commitments, mins = [], [], []
        for j in range(m):
            v = rng.next_u64() % (1 << (bit_length - 1))
            mins.append({"none": None, "third": v // 3, "eq": v}[strategy])
            blind = [O.random_not_zero(rng)] * extension_degree
            commitments.append(c.o_params.pc_gens.commit(v, blind))
            openings.append(O.CommitmentOpening(v, blind))
            if j == 0:
                c.expected_masks.append(list(blind) if m == 1 else None)
        w = O.RangeWitness(openings)
        sn = O.random_not_zero(rng) if m == 1 else None
        sp = O.RangeStatement(c.o_params, commitments, mins, sn)
        su = O.RangeStatement(c.o_params, commitments, mins, None)
        proof = O.prove_with_rng(M.Transcript(label), sp, w, rng)
        c.o_statements_private.append(sp)
        c.o_statements_public.append(su)
        c.o_proofs.append(proof)
        c.o_witnesses.append(w)
    return c


def attach_product(c, pkg, eng):
    """Product-side view of an oracle batch: everything as bytes, bound to device parameters."""
    c.params = pkg.RangeParameters.init(c.bit_length, c.m_max, pkg.create_pedersen_gens_with_extension_degree(c.t),
                                        engine=eng)
    c.statements_private, c.statements_public, c.proofs = [], [], []
    for sp in c.o_statements_private:
        comp = list(sp.commitments_compressed)
        c.statements_private.append(pkg.RangeStatement.init(c.params, comp, sp.minimum_value_promises,
                                                            sb(sp.seed_nonce) if sp.seed_nonce is not None else None))
        c.statements_public.append(pkg.RangeStatement.init(c.params, comp, sp.minimum_value_promises, None))
    for p in c.o_proofs:
        c.proofs.append(pkg.RangeProof.from_bytes(p.to_bytes()))
    c.transcripts = lambda: [pkg.Transcript.new(c.label) for _ in c.proofs]
    return c


def make_batch(pkg, eng, bit_length, aggregation, extension_degree, seed=b"8675309", strategy="third", m_max=None,
               label=LABEL):
    return attach_product(make_oracle_batch(bit_length, aggregation, extension_degree, seed, strategy, m_max, label),
                          pkg, eng)


def oracle_verify_trace(c, action=0, private=None, statements=None, proofs=None):
    """Run the oracle's verify() (no 256 cap) and return (masks as lists of 32-byte strings | None, trace)."""
    if statements is None:
        private = (action != 0) if private is None else private
        statements = c.o_statements_private if private else c.o_statements_public
    proofs = proofs if proofs is not None else c.o_proofs
    trace = {}
    masks = O.verify([M.Transcript(c.label) for _ in proofs], statements, proofs, action, trace=trace)
    out = [[sb(x) for x in m] if m is not None else None for m in masks]
    return out, trace


def trace_challenge_bytes(trace, max_rounds):
    """challenges in the layout of BPP_TRACE_CHALLENGES: per proof y, z, e_0.., e_final, zero padded to max_rounds+3"""
    out = b""
    for (y, z, rounds, e) in trace["challenges"]:
        row = [y, z] + list(rounds) + [e]
        row += [0] * (max_rounds + 3 - len(row))
        out += b"".join(sb(x) for x in row)
    return out
