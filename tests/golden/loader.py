"""Reader for tests/golden/bench_*.bin (pure data: proofs + statements; written by make_golden.py).

layout: "BPPB" u32 version, u32 n_proofs, u32 bit_length, u32 m, u32 t | u32 label_len, label |
        per proof: u32 proof_len, proof, m x 32 commitments, m x u64 min values, u8 seed_present, 32 seed bytes"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


def load_bench(name):
    raw = open(os.path.join(HERE, name), "rb").read()
    magic, ver, n, bits, m, t = struct.unpack_from("<4sIIIII", raw, 0)
    assert magic == b"BPPB" and ver == 1
    off = 24
    (ll,) = struct.unpack_from("<I", raw, off)
    off += 4
    label = raw[off:off + ll]
    off += ll
    items = []
    for _ in range(n):
        (pl,) = struct.unpack_from("<I", raw, off)
        off += 4
        proof = raw[off:off + pl]
        off += pl
        comm = [raw[off + 32 * j:off + 32 * j + 32] for j in range(m)]
        off += 32 * m
        mins = list(struct.unpack_from("<%dQ" % m, raw, off))
        off += 8 * m
        present = raw[off]
        seed = raw[off + 1:off + 33] if present else None
        off += 33
        items.append({"proof": proof, "commitments": comm, "min_values": mins, "seed_nonce": seed, "label": label})
    assert off == len(raw)
    return {"bit_length": bits, "m": m, "t": t, "label": label, "items": items}
