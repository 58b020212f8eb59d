"""Debug build of the library whose kernels fill their LDS with NaNs first (-DUHC_POISON_LDS): run the GPU tests with
UHC_LIB=uhc_amd/csrc/libuhc_amd_poison.so to catch reads of LDS the kernel has not written.  The same build checks guard words behind every LDS region
(-DUHC_GUARD_LDS) when the batch is created with UHC_GUARD_LDS=1: writes past a region's end -- "uhc guard: ..." lines on stderr."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import __graft_entry__ as g
CSRC = os.path.join(ROOT, "uhc_amd", "csrc")
g.compile_lib(lib=os.path.join(CSRC, "libuhc_amd_poison.so"), extra_flags=["-DUHC_POISON_LDS", "-DUHC_GUARD_LDS"], obj_dir=os.path.join(CSRC, "build_poison"))
