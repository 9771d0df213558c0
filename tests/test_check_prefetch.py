"""tools/check_prefetch.py is part of the build (raytracing_amd/csrc/Makefile): it must accept a correct pipeline and
reject anything that touches the destination SGPRs of a hand-pinned prefetch before its wait, on any path."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_prefetch", os.path.join(ROOT, "tools", "check_prefetch.py"))
cp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cp)

GOOD = """
kern:
	s_load_dwordx16 s[4:19], s[2:3], 0x0 ; PF_LOAD
.LBB0_1:
	s_waitcnt lgkmcnt(0) ; PF_WAIT
	s_load_dwordx16 s[36:51], s[0:1], 0x0 ; PF_LOAD
	v_pk_add_f32 v[0:1], s[4:5], v[2:3]
	v_cmp_lt_i32_e32 vcc, -1, v5
	s_cbranch_vccnz .LBB0_3
.LBB0_2:
	s_waitcnt lgkmcnt(0) ; PF_WAIT
	v_pk_add_f32 v[0:1], s[36:37], v[2:3]
	s_load_dwordx16 s[4:19], s[2:3], 0x0 ; PF_LOAD
	s_cbranch_scc1 .LBB0_1
	s_waitcnt lgkmcnt(0) ; PF_WAIT
	s_endpgm
.LBB0_3:
	v_cmp_gt_u32_e64 s[20:21], v15, v63
	s_branch .LBB0_2
"""


def run(tmp_path, text):
    f = tmp_path / "k.s"
    f.write_text(text)
    return cp.check(str(f))


def test_accepts_a_correct_pipeline(tmp_path):
    n, bad = run(tmp_path, GOOD)
    assert n == 3 and bad == []


def test_rejects_a_clobber_on_the_out_of_line_path(tmp_path):
    n, bad = run(tmp_path, GOOD.replace("v_cmp_gt_u32_e64 s[20:21]", "v_cmp_gt_u32_e64 s[36:37]"))
    assert any("s[36, 37]" in b[2] or "[36, 37]" in b[2] for b in bad)


def test_rejects_a_read_before_the_wait(tmp_path):
    n, bad = run(tmp_path, GOOD.replace("v_pk_add_f32 v[0:1], s[4:5], v[2:3]\n\tv_cmp", "v_pk_add_f32 v[0:1], s[40:41], v[2:3]\n\tv_cmp"))
    assert bad


def test_rejects_a_missing_wait(tmp_path):
    n, bad = run(tmp_path, GOOD.replace("	s_waitcnt lgkmcnt(0) ; PF_WAIT\n	s_endpgm", "	s_endpgm"))
    assert any("ends before" in b[2] for b in bad)


def test_the_built_library_passes():
    s = os.path.join(ROOT, "raytracing_amd", "csrc", "rt_lib.s")
    if os.path.exists(s):
        n, bad = cp.check(s)
        assert bad == []  # n == 0 when the build pins no prefetch by hand
