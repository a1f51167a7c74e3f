"""CPU suite: csrc/*_consts.inc are reproducible from first principles (tools/gen_constants.py)."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_constants_are_current(tmp_path):
    spec = importlib.util.spec_from_file_location("gen_constants", os.path.join(ROOT, "tools", "gen_constants.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    csrc = os.path.join(ROOT, "bulletproofs-plus_amd", "csrc")
    fld = open(os.path.join(csrc, "field_consts.inc")).read()
    scl = open(os.path.join(csrc, "scalar_consts.inc")).read()

    def arr(src, name):
        m = re.search(name + r"\[\d+\] = \{([^}]*)\}", src)
        return [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
    assert arr(fld, "FE_D") == g.limbs10(g.D)
    assert arr(fld, "FE_SQRT_M1") == g.limbs10(g.SQRT_M1)
    assert arr(fld, "FE_INVSQRT_A_MINUS_D") == g.limbs10(g.INVSQRT_A_MINUS_D)
    assert g.SC_R_BITS == 261  # nine 29-bit limbs (csrc/scalar.h)
    assert arr(scl, "SC_R1") == g.words8(pow(2, 261, g.L))
    assert arr(scl, "SC_R2") == g.words8(pow(2, 522, g.L))
    assert arr(scl, "SC_R3") == g.words8(pow(2, 783, g.L))
    assert arr(scl, "SC_P256") == g.words8(pow(2, 256, g.L))
    assert arr(scl, "SC_WIDE_HI") == g.words8(pow(2, 256 + 522, g.L))
    l29 = arr(scl, "SC_L29")
    assert sum(v << (29 * i) for i, v in enumerate(l29)) == g.L and l29[5:8] == [0, 0, 0] and l29[8] == 1 << 20
    linv = int(re.search(r"BPP_LINV29 0x([0-9a-f]+)u", scl).group(1), 16)
    assert (linv * g.L + 1) % 2**29 == 0
    pw = re.search(r"SC_POW2_R29\[64\]\[9\] = \{(.*?)\};", scl, re.S).group(1)
    rows = [[int(x.strip().rstrip("u"), 16) for x in r.split(",")] for r in re.findall(r"\{([^{}]*)\}", pw)]
    assert len(rows) == 64 and all(sum(v << (29 * i) for i, v in enumerate(r)) == (1 << e) * pow(2, 261, g.L) % g.L for e, r in enumerate(rows))
    l30 = [int(x.strip(), 16) for x in re.search(r"SC_L30\[9\] = \{([^}]*)\}", scl).group(1).split(",")]
    assert sum(v << (30 * i) for i, v in enumerate(l30)) == g.L
    linv30 = int(re.search(r"BPP_LINV30 0x([0-9a-f]+)u", scl).group(1), 16)
    assert (linv30 * g.L) % 2**30 == 1
