#!/bin/bash
# Register / spill / LDS metadata of every kernel in libuhc_amd.so (code-object notes): tools/kernel_meta.sh [lib.so]
set -e
LIB=${1:-$(dirname "$0")/../uhc_amd/csrc/libuhc_amd.so}
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$LIB" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | python3 -c '
import sys, re
cur = {}
def flush():
    if cur.get("name"):
        print("%-44s vgpr %3s agpr %3s sgpr %3s | spill vgpr %4s sgpr %4s | scratch %5s B | lds %6s B" % (cur["name"][:44], cur.get("vgpr_count"), cur.get("agpr_count"), cur.get("sgpr_count"), cur.get("vgpr_spill_count"), cur.get("sgpr_spill_count"), cur.get("private_segment_fixed_size"), cur.get("group_segment_fixed_size")))
for line in sys.stdin:
    m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "agpr_count" and cur.get("name") and "agpr_count" in cur: pass
    if k == "name" and not v.endswith(".kd") and "name" in cur and "vgpr_count" in cur:
        flush(); cur.clear()
    if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
        if k == "name" and (v.endswith(".kd") or k in cur and not v.startswith("_Z")): continue
        cur[k] = v
flush()
'
rm -rf $T
