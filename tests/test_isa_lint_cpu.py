"""CPU: tools/isa_lint.py on synthetic disassembly -- the pattern it exists for (a cross-lane read of a VGPR that was reloaded from scratch under a narrower EXEC mask,
DESIGN 4.1b "a hazard worth writing down") is reported, the safe shapes are not."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _lines(txt):
    return [(4 * k, l.strip()) for k, l in enumerate(txt.strip().splitlines())]


def test_reload_under_a_narrowed_mask_then_readlane_outside_is_a_hazard():
    import isa_lint
    haz, rev, st = isa_lint.lint_kernel("k", _lines("""
        v_cmp_gt_i32 vcc, 8, v0
        s_and_saveexec_b64 s[4:5], vcc
        scratch_load_dword v7, off, s32 offset:16
        v_add_f64 v[2:3], v[2:3], v[8:9]
        s_or_b64 exec, exec, s[4:5]
        v_readlane_b32 s6, v7, 40
    """))
    assert len(haz) == 1 and haz[0][3] == "v7" and not rev and st["reloads_under_narrow_exec"] == 1


def test_safe_shapes_are_not_reported():
    import isa_lint
    # reloaded with all lanes on; reloaded under a narrowed mask but overwritten before the read; readfirstlane inside the region of the reload
    for txt in ("""
        scratch_load_dword v7, off, s32 offset:16
        s_and_saveexec_b64 s[4:5], vcc
        v_readlane_b32 s6, v7, 3
        s_or_b64 exec, exec, s[4:5]
    """, """
        s_and_saveexec_b64 s[4:5], vcc
        scratch_load_dword v7, off, s32 offset:16
        s_or_b64 exec, exec, s[4:5]
        v_mov_b32 v7, v9
        v_readlane_b32 s6, v7, 40
    """, """
        s_and_saveexec_b64 s[4:5], vcc
        scratch_load_dword v7, off, s32 offset:16
        v_readfirstlane_b32 s6, v7
        s_or_b64 exec, exec, s[4:5]
    """):
        haz, rev, _ = isa_lint.lint_kernel("k", _lines(txt))
        assert not haz and not rev, txt


def test_readlane_inside_the_region_of_the_reload_is_listed_for_review():
    import isa_lint
    haz, rev, _ = isa_lint.lint_kernel("k", _lines("""
        s_and_saveexec_b64 s[4:5], vcc
        scratch_load_dword v7, off, s32 offset:16
        v_readlane_b32 s6, v7, 3
        s_or_b64 exec, exec, s[4:5]
    """))
    assert not haz and len(rev) == 1


def test_the_else_half_of_a_branch_is_another_mask():
    import isa_lint
    haz, _, _ = isa_lint.lint_kernel("k", _lines("""
        s_and_saveexec_b64 s[4:5], vcc
        scratch_load_dword v7, off, s32 offset:16
        s_xor_b64 exec, exec, s[4:5]
        v_readfirstlane_b32 s6, v7
        s_or_b64 exec, exec, s[4:5]
    """))
    assert len(haz) == 1  # the first active lane of the ELSE half is a lane the reload skipped
