"""CPU: libuhc_amd.so loads and exports every symbol include/uhc_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "uhc_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uhc_[a-z_0-9A-Z]+)\s*\(", src)))


def test_header_symbols_are_exported_and_listed():
    import __graft_entry__ as g
    g.build()
    from uhc_amd import _lib
    names = _declared()
    assert len(names) >= 20 and "uhc_env_step" in names and "uhc_batch_simulate" in names
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/uhc_amd.h but not exported"
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)
    # ... and nothing else: the launchers / accessors the translation units share (uhc_launch_*, uhc_internal_*) are not exports
    import subprocess
    dyn = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted({l.split()[-1] for l in dyn.splitlines() if " T " in l and l.split()[-1].startswith("uhc_")})
    assert exported == names, sorted(set(exported) ^ set(names))
    L.uhc_abi_version.restype = ctypes.c_int32
    header = open(os.path.join(ROOT, "include", "uhc_amd.h")).read()
    assert L.uhc_abi_version() == int(re.search(r"#define UHC_ABI_VERSION (\d+)", header).group(1))
    # the shipped library is a release build: no measurement switches (UHC_DEBUG bits 8-12 are compiled out and masked), no stage counters, no poison / guards
    L.uhc_build_flags.restype = ctypes.c_int32
    assert L.uhc_build_flags() == 0


def test_experiment_switches_exist_only_behind_the_build_flag():
    """UHC_DEBUG bits 8-12 (working-set fill, consumer cap, sticky tier 4) change which envs report windows / sweeps: every read of them in the kernels goes
    through UHC_EXP(bit), which is `false` unless the library is built with -DUHC_EXPERIMENTS, and the host masks the bits otherwise."""
    csrc = os.path.join(ROOT, "uhc_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".h", ".hip", ".cpp")):
            txt = re.sub(r"//.*", "", open(os.path.join(csrc, f)).read())
            for m in re.finditer(r"dbg\s*&\s*(0x[0-9a-fA-F]+|\d+)", txt):
                bit = int(m.group(1), 0)
                assert bit & 0x1f00 == 0 or (f == "uhc_capi.cpp" and "UHC_EXPERIMENTS" in txt[max(0, m.start() - 400):m.end() + 400]), (f, m.group(0))


def test_product_package_never_imports_the_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "uhc_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(d, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(d, f)
    for f in os.listdir(os.path.join(ROOT, "uhc_amd", "csrc")):
        if f.endswith((".hip", ".cpp", ".h")):
            assert "oracle" not in open(os.path.join(ROOT, "uhc_amd", "csrc", f)).read().replace("the oracle", "").replace("and the oracle", ""), f
