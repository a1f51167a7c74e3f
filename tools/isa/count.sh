#!/bin/bash
# VALU instruction mix per probe kernel (gfx950 ISA from hipcc -S); run from the repo root
set -e
cd "$(dirname "$0")"
SRC=${1:-fe_probe.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/fe_probe.s "$SRC"
python3 - <<'PY'
import re, collections
txt = open('/tmp/fe_probe.s').read()
for m in re.finditer(r'^(probe_\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ops = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line.startswith((';', '.', '/')) or line.endswith(':'):
            continue
        ops[line.split()[0]] += 1
    valu = sum(c for o, c in ops.items() if o.startswith('v_'))
    mad = sum(c for o, c in ops.items() if o.startswith(('v_mad_u64', 'v_mul_lo', 'v_mul_hi', 'v_mad_u32')))
    half = mad + sum(c for o, c in ops.items() if o.startswith(('v_lshlrev_b64', 'v_lshrrev_b64', 'v_ashrrev_i64')))
    vg = re.search(r'\.set %s\.num_vgpr, (\d+)' % name, txt)
    print('%-18s vgpr %3s valu %4d  mul/mad %4d  other %4d  | slots(mul=2) %5d | %s' % (name, vg.group(1) if vg else '?', valu, mad, valu - mad, valu + half,
          ' '.join('%s:%d' % (o, c) for o, c in ops.most_common(12) if o.startswith('v_'))))
PY
