"""Static instruction mix of the step kernels, by class, from the disassembly of the library's gfx950 code objects (which VALU instructions are
arithmetic and which are plumbing -- lane broadcasts, DPP, AGPR traffic, SGPR spills through v_writelane).  Static counts: every instruction once, however often
its loop runs; read beside the dynamic counters of profiles/*_pmc_VALU_F64*.txt.      python tools/isa_classes.py [lib.so] > profiles/rNN_isa_classes.txt"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "uhc_amd", "csrc", "libuhc_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin/"
T = tempfile.mkdtemp()
subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, T + "/fat.bin"])
d = open(T + "/fat.bin", "rb").read()
starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", d)] + [len(d)]
CLASSES = [
    ("f64 arithmetic (v_fma/fmac/mul/add/min/max/rcp/rsq/sqrt/div*_f64, v_cmp*_f64)", r"^v_(fma|fmac|mul|add|min|max|rcp|rsq|sqrt|div_fmas|div_fixup|div_scale|fract|trunc|floor|ceil|rndne|ldexp|frexp\w*|cmp\w*|cmpx\w*)_f64"),
    ("f64 <-> other conversions", r"^v_cvt_\w*f64|^v_cvt_f64"),
    ("lane broadcast: v_readlane / v_readfirstlane", r"^v_read(first)?lane"),
    ("SGPR spill traffic: v_writelane", r"^v_writelane"),
    ("DPP / permute (cross-lane reductions)", r"_dpp|^v_permlane|^ds_bpermute|^ds_permute|^ds_swizzle|^v_mov_b32_dpp"),
    ("AGPR traffic: v_accvgpr_read / write / mov", r"^v_accvgpr"),
    ("v_mov / v_cndmask / bit ops (b32, b64)", r"^v_(mov|cndmask|and|or|xor|not|bfe|bfi|lshl|lshr|ashr|lshlrev|lshrrev|ashrrev|perm|alignbit|lshl_or|and_or|or3|lshl_add|add_lshl)"),
    ("integer / address arithmetic (v_add/sub/mul/mad _u32/_i32/_co, v_cmp int)", r"^v_(add|sub|subrev|mul|mad|min|max|cmp\w*|cmpx\w*|addc|subb|add3|mad_u64)\w*_(u32|i32|u16|i16|u64|i64|co_u32|co_ci_u32|u32_u24|i32_i24|hi_u32|lo_u32)"),
    ("other VALU (f32, misc)", r"^v_"),
    ("LDS: ds_read / ds_write / ds_add ...", r"^ds_"),
    ("scratch (VGPR spill) loads / stores", r"^scratch_"),
    ("global / buffer / flat memory", r"^(global|buffer|flat)_"),
    ("scalar memory: s_load / s_buffer_load", r"^s_(load|buffer_load|store)"),
    ("s_waitcnt / s_nop / s_sleep / barriers", r"^s_(waitcnt|nop|sleep|barrier|sethalt|setprio)"),
    ("branches", r"^s_(cbranch|branch|setpc|swappc|call)"),
    ("other SALU", r"^s_"),
]
want = re.compile(r"uhc_step(_queue)?_kernelILi0E")
print(f"# static instruction mix of the control-step kernels (MODE 0) of {os.path.relpath(lib, ROOT)}, llvm-objdump -d of the gfx950 code objects; one column per kernel")
rows = collections.OrderedDict()
names = []
for k in range(len(starts) - 1):
    b = T + "/b%02d.bin" % k
    open(b, "wb").write(d[starts[k]:starts[k + 1]])
    co = b + ".co"
    if subprocess.call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + b, "--output=" + co, "--unbundle"], stderr=subprocess.DEVNULL) != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
        continue
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    cur = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1) if want.search(m.group(1)) else None
            if cur:
                t = re.search(r"uhc_step(_queue)?_kernelILi(\d)ELi(\d)ELb(\d)", cur)
                cur = ("queue " if t.group(1) else "") + f"<{t.group(2)}, {t.group(3)}, {'true' if t.group(4) == '1' else 'false'}>"
                if cur not in names:
                    names.append(cur)
            continue
        if not cur:
            continue
        ins = line.strip().split("//")[0].strip()
        if not ins or ins.startswith("."):
            continue
        op = ins.split()[0]
        full = ins
        for cname, rx in CLASSES:
            if re.search(rx, op) or (cname.startswith("DPP") and re.search(r"(row_|quad_perm|wave_|row_bcast|row_shr|row_newbcast)", full)):
                rows.setdefault(cname, collections.Counter())[cur] += 1
                break
print("class".ljust(84) + "".join(n.rjust(20) for n in names))
tot = collections.Counter(); valu = collections.Counter()
for cname, _ in CLASSES:
    c = rows.get(cname, {})
    print(cname.ljust(84) + "".join(str(c.get(n, 0)).rjust(20) for n in names))
    for n in names:
        tot[n] += c.get(n, 0)
        if CLASSES[[x[0] for x in CLASSES].index(cname)][1].startswith("^v_") or cname.startswith(("DPP", "lane", "SGPR", "AGPR", "f64")):
            valu[n] += c.get(n, 0)
print("all instructions".ljust(84) + "".join(str(tot[n]).rjust(20) for n in names))
print("VALU instructions".ljust(84) + "".join(str(valu[n]).rjust(20) for n in names))
f64 = rows.get(CLASSES[0][0], {})
print("f64 arithmetic / VALU".ljust(84) + "".join(f"{100.0 * f64.get(n, 0) / max(valu[n], 1):.1f} %".rjust(20) for n in names))
