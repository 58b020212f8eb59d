#!/bin/bash
# Register / spill / LDS metadata of EVERY kernel in libuhc_amd.so (code-object notes): tools/kernel_meta.sh [lib.so]
set -e
LIB=${1:-$(dirname "$0")/../uhc_amd/csrc/libuhc_amd.so}
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$LIB" $T/fat.bin
# one offload bundle per translation unit: split the section at the bundle magic, unbundle each code object
python3 - $T <<'PY'
import re, sys
d = open(sys.argv[1] + "/fat.bin", "rb").read()
starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", d)] + [len(d)]
for k in range(len(starts) - 1):
    open("%s/b%02d.bin" % (sys.argv[1], k), "wb").write(d[starts[k]:starts[k + 1]])
PY
: > $T/notes.txt
for f in $T/b*.bin; do
    /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$f --output=$f.co --unbundle
    /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f.co >> $T/notes.txt
done
python3 - $T/notes.txt <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
# one "- .agpr_count: ..." YAML item per kernel under amdhsa.kernels; keys are printed alphabetically, one per line
items = re.split(r"\n\s*- \.agpr_count:", txt)
rows = []
for it in items[1:]:
    it = ".agpr_count:" + it
    f = dict(re.findall(r"\.(\w+):\s+('?[\w$.@]+'?)", it))
    name = f.get("name", "?").strip("'")
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    rows.append((name, f))
for name, f in sorted(rows):
    print("%-46s vgpr %3s agpr %3s sgpr %3s | spill vgpr %4s sgpr %4s | scratch %5s B | static lds %6s B" % (
        name[:46], f.get("vgpr_count"), f.get("agpr_count"), f.get("sgpr_count"), f.get("vgpr_spill_count"), f.get("sgpr_spill_count"),
        f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size")))
PY
rm -rf $T
