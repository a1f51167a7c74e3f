"""CPU suite: the host packer of bpp_batch_upload (csrc/upload_host.h: pass A / pass B over untrusted proof and statement
bytes) and the shared arithmetic headers under AddressSanitizer + UBSan (sanitizers on the CPU build only).

Cases (csrc/hosttest_upload.cpp): reference error kinds of from_bytes on every truncation, random mutations, the
two-loop precedence of verify_statements_and_generators_consistency (src/range_proof.rs:637-682), the layout count ==
kernel count invariant for a proof with the largest accepted number of (L, R) pairs (BPP_MAX_WIRE_ROUNDS), refusal above that."""
import importlib
import os
import subprocess


def test_upload_packer_under_asan_ubsan():
    pkg = importlib.import_module("bulletproofs-plus_amd")
    exe = pkg._build.build_sanitizer_harness()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = r.stdout.split("\n")
    for case in ("valid_mixed", "precedence_degree_before_promise", "construction_error_first", "truncations", "mutations",
                 "max_wire_rounds_layout", "over_max_wire_rounds_refused", "bad_transcript_state", "null_proof",
                 "null_commitments", "merlin_equivalence_simple", "weight_chain_lockstep", "arithmetic_probes"):
        assert "ok " + case in lines, (case, r.stdout[-2000:])
    assert "all ok" in lines
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
