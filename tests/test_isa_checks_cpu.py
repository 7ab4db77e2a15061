"""build() refuses a library whose split-role kernels' ISA shows a hazard the compiler does not cover (__graft_entry__._check_isa: the static
checks tools/store_hazard_check.py and tools/mfma_hazard_check.py over the text hipcc --save-temps leaves behind).  Here: the checkers
themselves on small hand-written listings, and -- when the in-tree build has left its ISA text -- a re-run over the shipped kernels."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, text, tmp_path, name):
    p = tmp_path / name
    p.write_text(text)
    return subprocess.run([sys.executable, os.path.join(REPO, "tools", tool), str(p)], capture_output=True, text=True)


FUNC = "_Z4testv:\n"


def test_store_data_hazard_is_found_and_a_wait_state_clears_it(tmp_path):
    bad = FUNC + "\tbuffer_store_dwordx4 v[0:3], v4, s[0:3], s5 offen\n\tv_mov_b32_e32 v1, v9\n\ts_endpgm\n"
    r = _run("store_hazard_check.py", bad, tmp_path, "bad.s")
    assert r.returncode == 1 and "1 store-data hazard" in r.stdout, r.stdout
    two = FUNC + "\tbuffer_store_dwordx4 v[0:3], v4, s[0:3], s5 offen\n\ts_mov_b32 s9, 0\n\tv_mov_b32_e32 v3, v9\n\ts_endpgm\n"
    assert _run("store_hazard_check.py", two, tmp_path, "two.s").returncode == 1          # the second slot behind the store counts too
    ok = FUNC + "\tbuffer_store_dwordx4 v[0:3], v4, s[0:3], s5 offen\n\ts_nop 1\n\tv_mov_b32_e32 v1, v9\n\ts_endpgm\n"
    r = _run("store_hazard_check.py", ok, tmp_path, "ok.s")
    assert r.returncode == 0 and "0 store-data hazard" in r.stdout, r.stdout
    other = FUNC + "\tbuffer_store_dwordx4 v[0:3], v4, s[0:3], s5 offen\n\tv_mov_b32_e32 v7, v9\n\ts_endpgm\n"
    assert _run("store_hazard_check.py", other, tmp_path, "other.s").returncode == 0      # another register: no hazard
    narrow = FUNC + "\tbuffer_store_dwordx2 v[0:1], v4, s[0:3], s5 offen\n\tv_mov_b32_e32 v1, v9\n\ts_endpgm\n"
    assert _run("store_hazard_check.py", narrow, tmp_path, "narrow.s").returncode == 0    # 64-bit stores are not affected


def test_mfma_result_read_too_early_is_found(tmp_path):
    bad = FUNC + "\tv_mfma_f32_16x16x4_f32 v[0:3], a0, v8, 0\n\tv_add_f32_e32 v9, v0, v1\n\ts_endpgm\n"
    r = _run("mfma_hazard_check.py", bad, tmp_path, "mbad.s")
    assert r.returncode == 1, r.stdout
    ok = FUNC + "\tv_mfma_f32_16x16x4_f32 v[0:3], a0, v8, 0\n\ts_nop 7\n\ts_nop 4\n\tv_add_f32_e32 v9, v0, v1\n\ts_endpgm\n"
    r = _run("mfma_hazard_check.py", ok, tmp_path, "mok.s")
    assert r.returncode == 0, r.stdout


def test_the_shipped_split_role_kernels_pass_both_checks():
    isa = os.path.join(REPO, "neuraloc_amd", "csrc", "obj", "nocf_duo-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(isa):
        import pytest
        pytest.skip("no ISA text in the tree (build() leaves it behind when it compiles nocf_duo.hip)")
    for tool in ("store_hazard_check.py", "mfma_hazard_check.py"):
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", tool), isa], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
