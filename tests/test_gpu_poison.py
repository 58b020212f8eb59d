"""Uninitialised-LDS check: the parity tests once more against the debug build whose kernels fill their LDS with NaNs before they start
(tools/poison_build.py, -DUHC_POISON_LDS).  A kernel that reads LDS it has not written -- e.g. the tail of a chunked row load that meets
a zero multiplier -- passes or fails by what the previous workgroup on that CU left behind; with the poison it fails every time."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POISON_LIB = os.path.join(ROOT, "uhc_amd", "csrc", "libuhc_amd_poison.so")


@pytest.mark.skipif(not os.path.exists(POISON_LIB), reason="debug library not built (python tools/poison_build.py)")
def test_parity_holds_with_poisoned_lds():
    if os.environ.get("UHC_LIB"):
        pytest.skip("already running against an alternative library")
    import ctypes
    from uhc_amd import _lib
    dbg = ctypes.CDLL(POISON_LIB)
    if any(not hasattr(dbg, sym) for sym in _lib.SYMBOLS) or dbg.uhc_abi_version() != _lib.lib().uhc_abi_version():
        pytest.skip("debug library is older than the sources (python tools/poison_build.py)")
    # ... and with guard words behind every LDS region (UHC_GUARD_LDS=1 + the debug build's -DUHC_GUARD_LDS, round 5): a write past a region's end is
    # reported on stderr ("uhc guard: ... OVERWRITTEN") by uhc_batch_sync / uhc_batch_free
    env = dict(os.environ, UHC_LIB=POISON_LIB, UHC_GUARD_LDS="1")
    sel = ["tests/test_gpu_physics.py", "tests/test_gpu_selfcollision.py", "tests/test_gpu_ball.py", "tests/test_gpu_behaviour.py", "tests/test_gpu_env.py",
           "tests/test_gpu_env_objects.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-s", "--tb=short", "-m", "gpu"] + sel, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "uhc guard: LDS guard words on" in r.stdout, "the guard words were not laid out (UHC_GUARD_LDS)"
    hits = [ln for ln in r.stdout.splitlines() if "OVERWRITTEN" in ln]
    assert not hits, hits[:5]
    freed = [ln for ln in r.stdout.splitlines() if ln.startswith("uhc guard: batch of")]
    print(f"poisoned LDS + guard words: {len(freed)} batches created and freed in {len(sel)} test files, 0 guard words overwritten")
