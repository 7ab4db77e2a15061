"""CPU: host-side logic of the training / sharding path that needs no GPU."""
import torch

from neuraloc_amd.OCflow import costs_from_sums
from neuraloc_amd.distributed import allreduce_flat, reduce_cost_sums, shard_rows
from neuraloc_amd.train import _step_sizes


def test_step_sizes_follow_the_reference_time_bookkeeping():
    """src/OCflow.py:35,50 and :170: tk += h in double, each step re-derives h = (tk + h) - tk, cast to fp32"""
    for t0, t1, nt in [(0.0, 1.0, 80), (0.2, 0.7, 37), (0.0, 1.0, 3)]:
        hs = _step_sizes([t0, t1], nt)
        h = (t1 - t0) / nt
        tk = t0
        for k in range(nt):
            assert hs[k].item() == torch.tensor((tk + h) - tk, dtype=torch.float32).item()
            tk += h
        assert hs.dtype == torch.float32 and hs.shape == (nt,)


def test_costs_from_sums_on_host_tensors_matches_the_reference_formula():
    sums = torch.tensor([10.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 4.0])
    alph = [100.0, 9.0, 9.0, 0.5, 0.25, 0.125]
    Jc, cs = costs_from_sums(sums, alph)
    means = [v / 4.0 for v in (10.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0)]
    assert [float(c) for c in cs] == means
    assert float(Jc) == means[0] + 100.0 * means[1] + 0.5 * means[2] + 0.25 * means[3] + 0.125 * means[4]   # src/OCflow.py:88-90


def test_single_process_reductions_are_identities():
    ts = [torch.arange(6.0).reshape(2, 3), torch.ones(4)]
    out = allreduce_flat(ts)
    assert all(torch.equal(a, b) for a, b in zip(out, ts))
    s = torch.arange(8.0)
    assert torch.equal(reduce_cost_sums(s.clone()), s)
    assert shard_rows(10, 0, 1) == (0, 10)


# ---- per-shape JIT: which library a shape gets (neuraloc_amd/_lib.py lib_for); no GPU, no compute calls

def _fake_hipcc(tmp_path, ok=True, delay=0.0):
    """a stand-in compiler: copies the shipped library to the -o path (or fails), so the cache logic runs without hipcc's minute"""
    import stat
    from neuraloc_amd import _lib
    sh = tmp_path / "fake_hipcc.sh"
    body = "#!/bin/sh\nout=\nwhile [ $# -gt 0 ]; do if [ \"$1\" = -o ]; then out=$2; fi; shift; done\nsleep %g\n" % delay
    body += ("cp %s \"$out\"\n" % _lib.LIB_PATH) if ok else "echo 'error: no' >&2; exit 1\n"
    sh.write_text(body)
    sh.chmod(sh.stat().st_mode | stat.S_IEXEC)
    return str(sh)


def _wait_for(path, seconds=20.0):
    import os
    import time
    t0 = time.time()
    while time.time() - t0 < seconds:
        if os.path.exists(path):
            return True
        time.sleep(0.05)
    return False


def test_jit_modes_and_the_background_cache(tmp_path, monkeypatch):
    import os
    import shutil
    from neuraloc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import pytest
        pytest.skip("library not built")
    # a private copy of the library directory: the cache (csrc/jit/) of the real tree stays untouched
    csrc = tmp_path / "pkg" / "csrc"                            # <root>/<package>/csrc next to <root>/include, like the repo
    shutil.copytree(os.path.dirname(_lib.LIB_PATH), csrc, ignore=shutil.ignore_patterns("jit", "obj", "*stamps*"))
    os.makedirs(tmp_path / "include", exist_ok=True)
    shutil.copy(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(_lib.LIB_PATH))), "include", "nocf.h"), tmp_path / "include" / "nocf.h")
    monkeypatch.setattr(_lib, "LIB_PATH", str(csrc / "libnocf.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_jit_libs", {})
    monkeypatch.setattr(_lib, "_jit_started", set())
    monkeypatch.delenv("NOCF_LIB_PATH", raising=False)
    key = (8, 48, 2, 9, 4)
    so = _lib._jit_paths(key)[2]
    for v, want in (("0", "0"), ("", "0"), ("1", "1"), ("auto", "auto"), ("AUTO", "auto"), ("whatever", "auto")):
        monkeypatch.setenv("NOCF_JIT", v)
        assert _lib._jit_mode() == want
    monkeypatch.delenv("NOCF_JIT")
    assert _lib._jit_mode() == "auto"
    # off: the shipped library, nothing started
    monkeypatch.setenv("NOCF_JIT", "0")
    assert _lib.lib_for(*key) is _lib.lib() and not os.path.exists(os.path.dirname(so))
    # built-in shapes never compile
    monkeypatch.setenv("NOCF_JIT", "auto")
    monkeypatch.setenv("HIPCC", _fake_hipcc(tmp_path, ok=True, delay=0.3))
    assert _lib.lib_for(150, 512, 2, 10, 50) is _lib.lib() and not os.path.exists(os.path.dirname(so))
    # auto, nothing cached: generic now (and for the rest of the process), compilation in the background, one at a time
    first = _lib.lib_for(*key)
    assert first is _lib.lib()
    other = (8, 64, 2, 9, 4)
    assert _lib.lib_for(*other) is _lib.lib()                   # the lock is held: this one is not started
    assert _wait_for(so) and _wait_for(so + ".log")
    assert not _wait_for(_lib._jit_paths(other)[2], 0.5)
    assert _lib.lib_for(*key) is first                           # the same process keeps its choice
    # "the next process": a fresh decision table finds the cache
    monkeypatch.setattr(_lib, "_jit_libs", {})
    monkeypatch.setattr(_lib, "_jit_started", set())
    import time
    t0 = time.time()
    while os.path.exists(os.path.join(os.path.dirname(so), ".compiling")) and time.time() - t0 < 10:
        time.sleep(0.05)
    got = _lib.lib_for(*key)
    assert got is not _lib.lib() and got.nocf_version() == _lib.lib().nocf_version()
    # a failing compiler: log kept, no retry for the same sources, the rollout is never taken down
    monkeypatch.setenv("HIPCC", _fake_hipcc(tmp_path, ok=False))
    bad = (8, 80, 2, 9, 4)
    assert _lib.lib_for(*bad) is _lib.lib()
    assert _wait_for(_lib._jit_paths(bad)[2] + ".log") and not os.path.exists(_lib._jit_paths(bad)[2])
    t0 = time.time()
    while os.path.exists(os.path.join(os.path.dirname(so), ".compiling")) and time.time() - t0 < 10:
        time.sleep(0.05)
    monkeypatch.setattr(_lib, "_jit_libs", {})
    monkeypatch.setattr(_lib, "_jit_started", set())
    assert _lib._jit_start_background(bad) is False
    # no compiler at all: generic, silently
    monkeypatch.setenv("HIPCC", str(tmp_path / "does_not_exist"))
    monkeypatch.setattr(_lib, "_jit_libs", {})
    monkeypatch.setattr(_lib, "_jit_started", set())
    assert _lib.lib_for(8, 96, 2, 9, 4) is _lib.lib()


def test_contract_splits_the_rows_of_wide_products_and_keeps_the_sum():
    """neuraloc_amd/train.py _contract: wide outputs (the 512-wide network's weight gradients) are cut into row slabs, one batched
    GEMM, fixed-order sum; the result is X' Y whatever the slab count (host tensors take the same torch path)"""
    from neuraloc_amd.train import _contract
    g = torch.Generator().manual_seed(11)
    for K, m, n in ((16 * 8192, 130, 161), (8192 * 2 + 8, 200, 140), (1000, 160, 150)):      # 16 slabs; 8 slabs (K % 16 != 0); too short to split
        X = torch.randn(K, m, generator=g)
        Y = torch.randn(K, n, generator=g)
        want = (X.double().t() @ Y.double())
        got = _contract(X, Y)
        assert got.shape == (m, n)
        assert float((got.double() - want).abs().max()) <= 2e-5 * (K ** 0.5)
        acc = _contract(X, Y, got.clone())
        assert float((acc.double() - 2 * want).abs().max()) <= 4e-5 * (K ** 0.5)
        assert torch.equal(_contract(X, Y), got)                       # run-to-run identical


def test_shock_times_on_the_shared_step_grid():
    """neuraloc_amd.shock._on_shared_grid: which first segments (int(t_s nt) steps of t_s / int(t_s nt): src/plotter.py:815-818) are prefixes of
    ONE unshocked rollout to the largest shock time"""
    from neuraloc_amd.shock import _on_shared_grid
    times = [0.1 * k for k in range(1, 10)]                      # 0.30000000000000004 etc.: the values a sweep script makes
    T, N, h, on = _on_shared_grid((0.0, 1.0), 50, times)
    assert (N, on) == (45, [5, 10, 15, 20, 25, 30, 35, 40, 45]) and abs(T - 0.9) < 1e-15 and abs(h - 0.02) < 1e-15
    T, N, h, on = _on_shared_grid((0.0, 1.0), 50, [0.1, 0.33, 0.5, 0.999])
    assert on == [None, None, None, 49] or on[1] is None          # 0.33: 16 steps of 0.020625 -- not on the grid of the largest time
    T, N, h, on = _on_shared_grid((0.0, 1.0), 50, [0.2, 0.33, 0.5])
    assert (N, on) == (25, [10, None, 25])
    T, N, h, on = _on_shared_grid((0.0, 1.0), 20, [0.01, 0.5])   # int(0.01 * 20) = 0: no segment 1 at all (shock_rollout raises for it)
    assert on == [None, 10]
    assert _on_shared_grid((0.0, 1.0), 20, [0.01])[3] == [None]
