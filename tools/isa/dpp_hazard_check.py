#!/usr/bin/env python3
"""Guard against a gfx950 hazard the compiler does not cover: the result of a DPP instruction (v_mov_b32_dpp: the quad
broadcasts of the latency-form MSM kernels) read as store DATA by the very next instruction (seen with register spills:
`v_mov_b32_dpp v4, ..` / `scratch_store_dwordx2 off, v[4:5], ..`) arrives stale -- one wait state in between fixes it, a
wait state before the DPP instruction does not.  This script disassembles the gfx950 code object inside libbpp_hip.so and
fails if any kernel contains the pattern (DPP result consumed by an immediately following VMEM / scratch / LDS store or
permute).

Second check: no v_subrev / v_subbrev instruction in DPP form.  `v_subrev_u32_dpp d, A, B quad_perm:[3,3,3,3]` returns
B[lane 3] - A[lane] on the MI355X with this toolchain instead of B[lane] - A[lane 3] (sub and add in DPP form are fine);
the compiler's DPP combine produces it for `x - broadcast(y)`.  usage: dpp_hazard_check.py [path to libbpp_hip.so]"""
import os
import re
import subprocess
import sys
import tempfile

def llvm_bin(hipcc=None):
    """directory holding llvm-objdump / llvm-objcopy: next to the hipcc in use (<rocm>/bin/hipcc -> <rocm>/lib/llvm/bin), then
    the usual prefixes, then PATH; None if nowhere"""
    import shutil
    cands = []
    for h in (hipcc, os.environ.get("HIPCC"), shutil.which("hipcc")):
        if h and os.path.isabs(h):
            root = os.path.dirname(os.path.dirname(os.path.realpath(h)))
            cands += [os.path.join(root, "lib", "llvm", "bin"), os.path.join(root, "llvm", "bin")]
    cands += ["/opt/rocm/lib/llvm/bin", "/opt/rocm/llvm/bin"]
    w = shutil.which("llvm-objdump")
    if w:
        cands.append(os.path.dirname(w))
    for c in cands:
        if os.path.exists(os.path.join(c, "llvm-objdump")) and os.path.exists(os.path.join(c, "llvm-objcopy")):
            return c
    return None


LLVM = llvm_bin() or "/opt/rocm/lib/llvm/bin"
STORES = ("scratch_store", "global_store", "flat_store", "buffer_store", "ds_write", "ds_bpermute", "ds_permute", "ds_swizzle")


def reads(instr, reg):
    for t in instr.replace(",", " ").split()[1:]:
        if t == reg:
            return True
        m = re.match(r"v\[(\d+):(\d+)\]", t)
        if m and int(m.group(1)) <= int(reg[1:]) <= int(m.group(2)):
            return True
    return False


def check(so):
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so, os.devnull], check=True)
        data = open(fat, "rb").read()
        elf = os.path.join(d, "co.elf")
        open(elf, "wb").write(data[data.find(b"\x7fELF"):])
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", elf], check=True, capture_output=True,
                             text=True).stdout
    cur, prev, hits, n_dpp = None, None, [], 0
    for line in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur, prev = m.group(1), None
            continue
        p = line.strip().split("//")[0].strip()
        if not p or p.endswith(":"):
            continue
        if prev is not None and p.split()[0].startswith(STORES) and reads(p, prev):
            hits.append((cur, p))
        prev = None
        op = p.split()[0]
        if op.startswith(("v_subrev", "v_subbrev")) and ("_dpp" in op or " quad_perm:" in p or " row_" in p):
            hits.append((cur, p))
        if op.startswith("v_") and ("_dpp" in op or " quad_perm:" in p or " row_" in p):
            n_dpp += 1
            prev = p.replace(",", " ").split()[1]
    return n_dpp, hits


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "bulletproofs-plus_amd",
                                                            "libbpp_hip.so")
    n_dpp, hits = check(so)
    print("%d DPP instructions, %d hazards (result stored by the next instruction, or subrev in DPP form)" % (n_dpp, len(hits)))
    for fn, p in hits[:20]:
        print("  ", fn[:60], ":", p)
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
