"""Lint of the step kernels' gfx950 code for the miscompilation pattern round 5 met (DESIGN 4.1b "a hazard worth writing down"):

    v_readlane / v_readfirstlane reads ONE lane of a VGPR whatever EXEC says.  If that VGPR was last (re)loaded from scratch -- a spill reload -- while EXEC
    was NARROWER than the set of lanes the read may address, the lanes that were switched off at the reload still hold what they held before the spill
    slot was written: garbage.  (uhc_primal.h's first chain pass broadcast run coefficients with readlane inside `if (LANE < len)`; the 552-spill
    instantiation reloaded the register inside the branch and produced NaNs, the 6-spill one did not.)

What is checked, per kernel, on a linear walk of the disassembly (llvm-objdump -d of the library's code objects):
  * EXEC-narrowing regions: s_and_saveexec_b64 / s_andn2_saveexec / s_and_b64 exec / s_andn2_b64 exec open one (depth + 1), s_or_b64 exec, exec, sN /
    s_mov_b64 exec, sN / s_or_saveexec close it (depth - 1, never below 0); branch targets reset the walk's depth to what it was when the branch was seen;
  * a VGPR written by scratch_load_* at depth > 0 is TAINTED (with the depth and the instruction's address); any other write to it clears the taint;
  * v_readlane_b32 of a tainted VGPR is a HIT; v_readfirstlane_b32 (first ACTIVE lane) is a hit only when it executes at a depth SMALLER than the reload's
    (EXEC has widened again since: the first active lane may be one the reload skipped).
The walk is linear, not a data-flow analysis over the CFG: it can miss a path and it can flag a reload whose narrower EXEC provably contains the lane read.
A hit is therefore a place to LOOK AT, listed with kernel, address and the reload it pairs with; `--strict` makes any hit an error (exit 1), which
__graft_entry__.build() turns on with UHC_LINT_STRICT=1.

    python tools/isa_lint.py [lib.so] [--strict] [--all-kernels]      > profiles/rNN_isa_lint.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def code_objects(lib, tmp):
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, tmp + "/fat.bin"])
    d = open(tmp + "/fat.bin", "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", d)] + [len(d)]
    for k in range(len(starts) - 1):
        b = tmp + "/b%02d.bin" % k
        open(b, "wb").write(d[starts[k]:starts[k + 1]])
        co = b + ".co"
        rc = subprocess.call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + b, "--output=" + co, "--unbundle"],
                             stderr=subprocess.DEVNULL)
        if rc == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            yield co


def vregs(tok):
    """'v12' -> [12]; 'v[4:7]' -> [4, 5, 6, 7]; anything else -> []"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


class Exec:
    """One value of the EXEC mask on the walk: `parent` is the state it was narrowed from (None: the kernel's entry mask), so that `a.within(b)` says that
    every lane on in a is on in b."""
    _n = 0

    def __init__(self, parent):
        Exec._n += 1
        self.id, self.parent = Exec._n, parent

    def within(self, other):
        s = self
        while s is not None:
            if s is other:
                return True
            s = s.parent
        return False


SAVE = re.compile(r"^s_(and|andn2|andn1|or|xor|nand|nor|xnor)_saveexec_b64\s+(s\[\d+:\d+\]|vcc)")
NARROW = re.compile(r"^s_(and|andn2)_b64\s+exec,\s*exec,")
RESTORE = re.compile(r"^s_(or|mov|xor)_b64\s+exec,\s*(?:exec,\s*)?(s\[\d+:\d+\]|vcc|-1)")
COPY = re.compile(r"^s_mov_b64\s+(s\[\d+:\d+\]|vcc),\s*exec")


def lint_kernel(name, lines):
    """-> (hazards, reviews, stats).  hazard: a cross-lane read in an EXEC state that is NOT within the reload's (EXEC widened or changed since the reload: lanes the
    reload skipped can be addressed); review: v_readlane (explicit lane) inside the very region of the reload -- safe only if the lane index stays inside the mask."""
    hazards, reviews = [], []
    root = Exec(None)
    cur = root
    saved = {}     # sgpr pair -> the Exec state it holds
    taint = {}     # vgpr -> (Exec state at the reload, address of the reload)
    state_at = {}  # branch target address -> Exec state when the branch was seen
    stats = collections.Counter()
    for addr, ins in lines:
        if addr in state_at:
            cur = state_at.pop(addr)
        op = ins.split()[0]
        args = [a.strip() for a in ins[len(op):].split(",")]
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-f]+)>|\b0x([0-9a-f]+)\b", ins)
            if m:
                state_at.setdefault(int(m.group(1) or m.group(2), 16), cur)
            continue
        m = SAVE.search(ins)
        if m:
            saved[m.group(2)] = cur
            cur = Exec(cur) if m.group(1) in ("and", "andn2", "andn1") else Exec(None)  # (or / xor / ... saveexec: not a subset of the old mask: an unrelated state)
            stats["regions"] += 1
            continue
        m = COPY.search(ins)
        if m:
            saved[m.group(1)] = cur
            continue
        if NARROW.search(ins):
            cur = Exec(cur)
            stats["regions"] += 1
            continue
        m = RESTORE.search(ins)
        if m:
            src = m.group(2)
            if src == "-1":
                cur = root
            elif m.group(1) == "xor":
                cur = Exec(saved.get(src).parent if saved.get(src) is not None else None)  # the else half of an if / else: the complement inside the enclosing mask
            else:
                cur = saved.get(src) or Exec(None)  # (a mask from somewhere the walk did not see: unrelated to everything)
            continue
        if re.match(r"^s_\w+\s+exec\b", ins):
            cur = Exec(None)  # any other write to EXEC: unknown
            continue
        if op.startswith("scratch_load"):
            stats["scratch_loads"] += 1
            for v in vregs(args[0]):
                if cur is not root:
                    taint[v] = (cur, addr)
                    stats["reloads_under_narrow_exec"] += 1
                else:
                    taint.pop(v, None)
            continue
        if op in ("v_readlane_b32", "v_readfirstlane_b32"):
            stats[op] += 1
            for v in vregs(args[1]):
                if v in taint:
                    st0, a0 = taint[v]
                    if not cur.within(st0):
                        hazards.append((name, addr, op, f"v{v}", a0))
                    elif op == "v_readlane_b32":
                        reviews.append((name, addr, op, f"v{v}", a0))
            continue
        if op.startswith(("v_", "ds_read", "global_load", "buffer_load", "flat_load")):
            for v in vregs(args[0]):  # destination: a fresh value in the lanes EXEC has on -- which lanes a later cross-lane read addresses is the compiler's business again
                taint.pop(v, None)
    return hazards, reviews, stats


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = argv[0] if argv else os.path.join(ROOT, "uhc_amd", "csrc", "libuhc_amd.so")
    strict = "--strict" in sys.argv
    want = re.compile(r"uhc_" if "--all-kernels" in sys.argv else r"uhc_step(_queue)?_kernel")
    tmp = tempfile.mkdtemp()
    all_hits, all_rev = [], []
    print(f"# tools/isa_lint.py {os.path.relpath(lib, ROOT)}: cross-lane reads (v_readlane / v_readfirstlane) of VGPRs reloaded from scratch under a narrowed EXEC mask")
    print(f"# {'kernel':58s} {'instr':>7s} {'regions':>8s} {'scratch_load':>12s} {'under narrow EXEC':>18s} {'readlane':>9s} {'readfirstlane':>13s} {'HAZ':>5s} {'review':>7s}")
    for co in code_objects(lib, tmp):
        dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        cur, lines = None, []
        kernels = []
        for line in dis.splitlines():
            m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
            if m:
                if cur and lines:
                    kernels.append((cur, lines))
                cur, lines = (m.group(2) if want.search(m.group(2)) else None), []
                continue
            if not cur:
                continue
            m = re.match(r"^\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
            if not m or not m.group(1):
                continue
            lines.append((int(m.group(2), 16), m.group(1).strip()))
        if cur and lines:
            kernels.append((cur, lines))
        for kname, klines in kernels:
            short = subprocess.run(["c++filt", kname], capture_output=True, text=True).stdout.strip() or kname
            short = re.sub(r"\(.*", "", short).replace("void ", "")
            hits, rev, st = lint_kernel(short, klines)
            all_hits += hits
            all_rev += rev
            print(f"  {short[:58]:58s} {len(klines):7d} {st['regions']:8d} {st['scratch_loads']:12d} {st['reloads_under_narrow_exec']:18d} {st['v_readlane_b32']:9d} {st['v_readfirstlane_b32']:13d} {len(hits):5d} {len(rev):7d}")
    for name, addr, op, v, a0 in all_hits:
        print(f"HAZARD {name}: {op} of {v} at 0x{addr:x} executes under an EXEC mask that is not inside the one {v} was reloaded from scratch under (0x{a0:x})")
    by = collections.Counter((n, v, a0) for n, _, _, v, a0 in all_rev)
    for (n, v, a0), k in sorted(by.items()):
        print(f"review {n}: {k} v_readlane of {v} inside the narrowed region it was reloaded in (0x{a0:x}): safe iff the lane index stays inside that mask")
    print(f"# {len(all_hits)} hazard(s), {len(all_rev)} read(s) to review")
    if strict and all_hits:
        sys.exit(1)


if __name__ == "__main__":
    main()
