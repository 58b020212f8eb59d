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


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(POISON_LIB), reason="debug library not built (python tools/poison_build.py)")
def test_guard_words_report_an_overwrite():
    """The guard mechanism itself: UHC_GUARD_LDS=2 declares the first two doubles of the fast tier's qpos a guard word; the first control step writes
    them, and uhc_batch_sync must say so -- an overwritten guard word cannot go unnoticed in the run above."""
    code = ("import numpy as np, torch\n"
            "from uhc_amd import sim as S\n"
            "m = S.load_asset_model(); c = S.make_ctrl(m)\n"
            "z = np.load('uhc_amd/assets/standing_neutral.npz')\n"
            "b = S.SimBatch(m, c, 4)\n"
            "b.set_state(torch.from_numpy(np.tile(z['qpos'], (4, 1))), torch.zeros(4, m.nv, dtype=torch.float64))\n"
            "b.simulate(torch.zeros(4, c.action_dim, dtype=torch.float64, device='cuda'), torch.zeros(4, m.nu, dtype=torch.float64, device='cuda'))\n"
            "b.sync(); b.close()\n")
    out = {}
    for mode in ("1", "2"):
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, UHC_LIB=POISON_LIB, UHC_GUARD_LDS=mode),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:]
        out[mode] = r.stdout
    if "uhc guard: LDS guard words on" not in out["1"]:
        pytest.skip("debug library is older than the guard words (python tools/poison_build.py)")
    assert "OVERWRITTEN" not in out["1"], out["1"][-2000:]
    assert "OVERWRITTEN" in out["2"] and "tier 1, persistent region" in out["2"], out["2"][-2000:]

