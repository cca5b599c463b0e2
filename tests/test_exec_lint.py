"""tools/exec_lint.py (the build's check for vector code placed ahead of a block's EXEC restore) on synthetic assembly."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import exec_lint

HEAD = """_ZN4klhr4testEv:
	s_andn2_b64 exec, exec, s[38:39]
	s_cbranch_execz .LBB0_7
; %bb.6:
	v_add_f64 v[0:1], v[0:1], v[2:3]
	s_branch .LBB0_9
"""


def _run(tmp_path, body):
    f = tmp_path / "k.s"
    f.write_text(HEAD + body)
    return exec_lint.lint(str(f))


def test_copies_ahead_of_the_restore_are_flagged(tmp_path):
    bad = _run(tmp_path, ".LBB0_7:\n\tv_accvgpr_write_b32 a48, v192\n\ts_mov_b32 s77, 0x40900000\n\ts_or_b64 exec, exec, s[38:39]\n\tv_mov_b32_e32 v1, v2\n")
    assert len(bad) == 1 and bad[0][1] == ".LBB0_7" and bad[0][3] == 1


def test_restore_first_is_clean(tmp_path):
    assert _run(tmp_path, ".LBB0_7:\n\ts_or_b64 exec, exec, s[38:39]\n\tv_accvgpr_write_b32 a48, v192\n") == []


def test_lane_moves_ignore_exec(tmp_path):
    # SGPR spill traffic (v_writelane / v_readlane) does not depend on EXEC and may sit ahead of the restore
    assert _run(tmp_path, ".LBB0_7:\n\tv_writelane_b32 v254, s6, 30\n\tv_readlane_b32 s4, v253, 7\n\ts_or_b64 exec, exec, s[38:39]\n") == []


def test_fallthrough_out_of_a_divergent_loop(tmp_path):
    body = ".LBB0_7:\n\ts_or_b64 exec, exec, s[38:39]\n.LBB0_8:\n\tv_add_f64 v[0:1], v[0:1], 1.0\n\ts_andn2_b64 exec, exec, s[4:5]\n\ts_cbranch_execnz .LBB0_8\n; %bb.9:\n\tv_accvgpr_read_b32 v0, a24\n\ts_or_b64 exec, exec, s[4:5]\n"
    bad = _run(tmp_path, body)
    assert len(bad) == 1 and bad[0][3] == 1
