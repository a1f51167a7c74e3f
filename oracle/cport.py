"""ORACLE (test infrastructure / CPU baseline only): ctypes view of oracle/c/liboracle.so.

Used by tests/ (C port vs pyref, large fixtures), by __graft_entry__.smoke() and by bench.py's cpu_baseline leg.
Never imported by the product package."""
import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, byref, c_char_p, c_double, c_int, c_size_t, c_uint8, c_uint32, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "c", "liboracle.so")


class Item(Structure):
    _fields_ = [("proof", c_void_p), ("proof_len", c_size_t), ("commitments32", c_void_p), ("m", c_uint32),
                ("min_values", c_void_p), ("min_present", c_void_p), ("seed_nonce32", c_void_p),
                ("transcript_label", c_void_p), ("label_len", c_size_t)]


class Trace(Structure):
    _fields_ = [("challenges", c_void_p), ("rmax", c_uint32), ("rng_out", c_void_p), ("weights", c_void_p),
                ("static_scalars", c_void_p), ("dynamic_scalars", c_void_p), ("msm_result", c_void_p)]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "c")], check=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = ctypes.CDLL(LIB)
        L.oracle_params_new.restype = c_void_p
        L.oracle_params_new.argtypes = [c_uint32, c_uint32, c_uint32]
        L.oracle_params_free.argtypes = [c_void_p]
        L.oracle_params_export.argtypes = [c_void_p] * 5
        L.oracle_commit.argtypes = [c_void_p, c_uint64, c_void_p, c_uint32, c_void_p]
        L.oracle_prove.argtypes = [c_void_p, c_void_p, c_size_t, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p, POINTER(c_size_t), c_void_p]
        L.oracle_verify.argtypes = [c_void_p, POINTER(Item), c_size_t, c_int, c_void_p, c_void_p, POINTER(Trace)]
        L.oracle_verify_timed.argtypes = [c_void_p, POINTER(Item), c_size_t, c_size_t, c_int, POINTER(c_double)]
        L.oracle_verify_timed_mt.argtypes = [c_void_p, POINTER(Item), c_size_t, c_size_t, c_int, c_int, POINTER(c_double)]
        L.oracle_verify_action_timed_mt.argtypes = [c_void_p, POINTER(Item), c_size_t, c_size_t, c_int, c_int, c_int, POINTER(c_double)]
        L.oracle_prove_timed_mt.argtypes = [c_void_p, c_void_p, c_size_t, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_size_t, c_size_t, c_int, c_int, c_void_p, c_size_t, POINTER(c_double)]
        L.oracle_keccak_count.restype = c_uint64
        L.oracle_nonce.argtypes = [c_void_p, c_char_p, c_int, c_int, c_void_p]
        L.oracle_msm.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
        _lib = L
    return _lib


def _b(data):
    return (c_uint8 * max(len(data), 1)).from_buffer_copy(data if len(data) else b"\0")


class Params:
    def __init__(self, n, m_max, t):
        self.n, self.m_max, self.t = n, m_max, t
        self.h = lib().oracle_params_new(n, m_max, t)
        if not self.h:
            raise ValueError("bad parameters")

    def export(self):
        nm = self.n * self.m_max
        gi, hi = (c_uint8 * (32 * nm))(), (c_uint8 * (32 * nm))()
        h, g = (c_uint8 * 32)(), (c_uint8 * (32 * self.t))()
        lib().oracle_params_export(self.h, gi, hi, h, g)
        split = lambda raw, k: [bytes(raw)[32 * i:32 * i + 32] for i in range(k)]
        return split(gi, nm), split(hi, nm), bytes(h), split(g, self.t)

    def commit(self, v, blindings):
        out = (c_uint8 * 32)()
        rc = lib().oracle_commit(self.h, v, _b(b"".join(blindings)), len(blindings), out)
        if rc:
            raise ValueError("commit rc=%d" % rc)
        return bytes(out)

    def prove(self, label, values, blindings, min_values, seed_nonce, ext_rng):
        """values: list[int]; blindings: list[list[bytes32]] (m x t); min_values: list[int|None];
        seed_nonce: bytes32|None; ext_rng: bytes, (rounds+3)*32 -> (proof bytes, compressed commitments)"""
        m = len(values)
        vals = (c_uint64 * m)(*values)
        mv = (c_uint64 * m)(*[(x if x is not None else 0) for x in min_values])
        mp = (c_uint8 * m)(*[(1 if x is not None else 0) for x in min_values])
        out = (c_uint8 * 4096)()
        n = c_size_t()
        comm = (c_uint8 * (32 * m))()
        rc = lib().oracle_prove(self.h, _b(label), len(label), m, vals, _b(b"".join(b"".join(b) for b in blindings)),
                                mv, mp, _b(seed_nonce) if seed_nonce else None, _b(ext_rng), len(ext_rng), out,
                                byref(n), comm)
        if rc:
            raise ValueError("prove rc=%d" % rc)
        return bytes(out)[:n.value], [bytes(comm)[32 * j:32 * j + 32] for j in range(m)]

    def items(self, batch):
        """batch: list of dicts {proof, commitments(list[bytes32]), min_values(list[int|None]), seed_nonce, label}"""
        keep = []
        arr = (Item * len(batch))()
        for i, b in enumerate(batch):
            m = len(b["commitments"])
            pb, cb = _b(b["proof"]), _b(b"".join(b["commitments"]))
            mv = (c_uint64 * m)(*[(x if x is not None else 0) for x in b["min_values"]])
            mp = (c_uint8 * m)(*[(1 if x is not None else 0) for x in b["min_values"]])
            lb = _b(b["label"])
            keep += [pb, cb, mv, mp, lb]
            it = arr[i]
            it.proof, it.proof_len = ctypes.cast(pb, c_void_p), len(b["proof"])
            it.commitments32, it.m = ctypes.cast(cb, c_void_p), m
            it.min_values, it.min_present = ctypes.cast(mv, c_void_p), ctypes.cast(mp, c_void_p)
            if b.get("seed_nonce"):
                sb = _b(b["seed_nonce"])
                keep.append(sb)
                it.seed_nonce32 = ctypes.cast(sb, c_void_p)
            it.transcript_label, it.label_len = ctypes.cast(lb, c_void_p), len(b["label"])
        return arr, keep

    def verify(self, batch, action=0, want_trace=False):
        """-> (rc, masks list[list[bytes32]|None], trace dict|None); rc uses the ProofError numbering"""
        arr, keep = self.items(batch)
        n, t = len(batch), self.t
        masks, present = (c_uint8 * (32 * t * n))(), (c_uint8 * n)()
        tr = None
        bufs = {}
        if want_trace:
            rounds = [((len(b["proof"]) - 1) // 32 - t - 5) // 2 for b in batch]
            rmax = max(rounds)
            max_mn = max(len(b["commitments"]) for b in batch) * self.n
            total_dyn = sum(len(b["commitments"]) + 3 + 2 * r for b, r in zip(batch, rounds))
            bufs = {"challenges": (c_uint8 * (n * (rmax + 3) * 32))(), "rng_out": (c_uint8 * (n * 32))(),
                    "weights": (c_uint8 * (n * 32))(), "static_scalars": (c_uint8 * ((2 * max_mn + t + 1) * 32))(),
                    "dynamic_scalars": (c_uint8 * (total_dyn * 32))(), "msm_result": (c_uint8 * 32)()}
            tr = Trace(ctypes.cast(bufs["challenges"], c_void_p), rmax, ctypes.cast(bufs["rng_out"], c_void_p),
                       ctypes.cast(bufs["weights"], c_void_p), ctypes.cast(bufs["static_scalars"], c_void_p),
                       ctypes.cast(bufs["dynamic_scalars"], c_void_p), ctypes.cast(bufs["msm_result"], c_void_p))
        rc = lib().oracle_verify(self.h, arr, n, action, masks, present, byref(tr) if tr is not None else None)
        raw = bytes(masks)
        out = [[raw[(i * t + k) * 32:(i * t + k) * 32 + 32] for k in range(t)] if present[i] else None
               for i in range(n)]
        return rc, out, ({k: bytes(v) for k, v in bufs.items()} if want_trace else None)

    def verify_timed(self, batch, chunk, iters):
        arr, keep = self.items(batch)
        sec = c_double()
        rc = lib().oracle_verify_timed(self.h, arr, len(batch), chunk, iters, byref(sec))
        return rc, sec.value

    def verify_timed_mt(self, batch, chunk, iters, threads):
        """`threads` host threads, thread k verifying the k-th `chunk`-proof slice of `batch` `iters` times -> (rc, wall s)"""
        arr, keep = self.items(batch)
        sec = c_double()
        rc = lib().oracle_verify_timed_mt(self.h, arr, len(batch), chunk, iters, threads, byref(sec))
        return rc, sec.value

    def verify_action_timed_mt(self, batch, chunk, action, iters, threads):
        """as verify_timed_mt under any VerifyAction (masks computed and dropped) -> (rc, wall s)"""
        arr, keep = self.items(batch)
        sec = c_double()
        rc = lib().oracle_verify_action_timed_mt(self.h, arr, len(batch), chunk, action, iters, threads, byref(sec))
        return rc, sec.value

    def prove_timed_mt(self, label, values, blindings, min_values, min_present, seeds, ext, iters, threads, proof_len=0):
        """numpy arrays in bench.make_inputs' layout (values n x m u64, blindings n x m x t x 32 u8, min_values n x m u64, min_present
        n x m u8, seeds n x 32 u8 or None, ext n x ext_len u8): thread k proves items k, k + threads, ... `iters` times
        -> (rc, wall s, proofs n x proof_len u8 array or None)"""
        import numpy as np
        n, m = values.shape
        values, blindings = np.ascontiguousarray(values, dtype=np.uint64), np.ascontiguousarray(blindings, dtype=np.uint8)
        min_values, min_present = np.ascontiguousarray(min_values, dtype=np.uint64), np.ascontiguousarray(min_present, dtype=np.uint8)
        ext = np.ascontiguousarray(ext, dtype=np.uint8)
        seeds = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint8)
        out = np.zeros((n, proof_len), dtype=np.uint8) if proof_len else None
        sec = c_double()
        ptr = lambda a: None if a is None else a.ctypes.data_as(c_void_p)
        rc = lib().oracle_prove_timed_mt(self.h, _b(label), len(label), m, ptr(values), ptr(blindings), ptr(min_values), ptr(min_present),
                                         ptr(seeds), ptr(ext), ext.shape[1], n, iters, threads, ptr(out), proof_len, byref(sec))
        return rc, sec.value, out

    def close(self):
        if self.h:
            lib().oracle_params_free(self.h)
            self.h = None
