#!/bin/bash
# VGPR / SGPR / scratch (spill) / LDS per kernel of the built libbpp_hip.so (gfx950 code object notes); no GPU needed
set -e
LIB=${1:-$(dirname "$0")/../../bulletproofs-plus_amd/libbpp_hip.so}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$LIB" --output="$TMP/co.o" --unbundle 2>/dev/null || \
  /opt/rocm/bin/roc-obj-extract "$LIB" >/dev/null 2>&1 || true
if [ ! -s "$TMP/co.o" ]; then
  # fall back: the fat binary section
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$TMP/fat.bin" "$LIB"
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/fat.bin" --output="$TMP/co.o" --unbundle
fi
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/co.o" | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    print("%-34s vgpr %4s sgpr %4s scratch %5s lds %6s wg %5s" % (g("name"), g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("max_flat_workgroup_size")))
'
rm -rf "$TMP"
