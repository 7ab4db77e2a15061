"""CPU: the C-ABI library builds, loads, and exports every symbol include/nocf.h declares;
argument errors come back as codes (no compute call needs a GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

import __graft_entry__ as entry
from neuraloc_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    entry.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(L):
    hdr = open(os.path.join(REPO, "include", "nocf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)           # comments stripped: a prototype inside a doc block is not a declaration
    hdr = re.sub(r"//[^\n]*", " ", hdr)
    names = set(re.findall(r"\b(nocf_[a-z0-9_]+)\s*\(", hdr))
    assert {"nocf_rollout_f32", "nocf_phi_grad_f32", "nocf_phi_forward_f32", "nocf_prob_eval_f32",
            "nocf_version", "nocf_workspace_bytes", "nocf_ctrl_dim", "nocf_selftest_mfma"} <= names
    for n in names:
        assert hasattr(L, n), f"{n} declared in nocf.h but not exported"
    assert L.nocf_version() == 113
    # ... and every export of the library that the Python layer binds is declared (a C caller sees the same surface)
    bound = set(re.findall(r"\bL\.(nocf_[a-z0-9_]+)\b", open(os.path.join(REPO, "neuraloc_amd", "_lib.py")).read()))
    assert bound <= names, f"bound but not declared in nocf.h: {sorted(bound - names)}"


def test_header_compiles_as_c_and_declares_what_it_documents(tmp_path):
    """nocf.h is a C header: a C translation unit that includes it and takes the address of every documented entry point compiles
    (gcc -fsyntax-only).  Round 4 shipped two prototypes INSIDE a comment block; the regex scan above could not see that."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this box")
    raw = open(os.path.join(REPO, "include", "nocf.h")).read()
    documented = set(re.findall(r"\b(nocf_[a-z0-9_]+)\s*\(", raw))      # comments included: everything the header talks about as a call
    documented = {n for n in documented if hasattr(_lib.lib(), n)}
    src = tmp_path / "use_nocf.c"
    body = "\n".join(f"    p[{i}] = (void*)&{n};" for i, n in enumerate(sorted(documented)))
    src.write_text('#include "nocf.h"\nvoid take(void** p) {\n' + body + "\n}\n")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(REPO, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_workspace_sizes(L):
    # swarm50: opening 8*40*64 + closing 3*128*64 + (forward, backward) 2*8*128*64 float4 images + padded vectors
    n4 = 8 * 40 * 64 + 3 * 128 * 64 + 2 * 8 * 128 * 64
    nv = 8 * 64 + 8 * 64 + 8 * 64 + 3 * 64 + 10 * 151 + 2          # ... + the copy of A (padded to 4 floats)
    got = L.nocf_workspace_bytes(150, 512, 2)
    assert (n4 * 4 + nv) * 4 < got <= (n4 * 4 + nv) * 4 + 1024          # + the plan record
    assert L.nocf_workspace_bytes(4, 32, 2) > 0
    assert L.nocf_workspace_bytes(4, 32, 1) == 0          # nTh < 2 is rejected (src/Phi.py:25-27)


def test_argument_errors_are_codes(L):
    assert L.nocf_rollout_f32(None, None, None, 1, 0.0, 1.0, 1, 4, None, None, None, None, None, None, None, 0, None) == -1
    pb = _lib.NocfProb()
    pb.kind, pb.n_agents = 2, 3
    assert L.nocf_ctrl_dim(C.byref(pb), 36) == 12
    pb.kind = 0
    assert L.nocf_ctrl_dim(C.byref(pb), 36) == 36


def test_double_precision_and_column_sum_entries_check_their_arguments(L):
    """nocf_rollout_f64 / nocf_phi_f64 / nocf_prob_eval_f64 / nocf_colsum_f32 return codes for bad arguments (nothing is launched)"""
    for n in ("nocf_rollout_f64", "nocf_phi_f64", "nocf_prob_eval_f64", "nocf_workspace_bytes_f64", "nocf_colsum_f32"):
        assert hasattr(L, n), n
    # workspace: K0^T image + (nTh-1) transposed layers + the packed operand images of the matrix-pipe products (opening 512 x 151:
    # 4 waves x 38 k-steps x 8 groups x 64; two 512 x 512 layer images; closing 151 x 512: 4 x 128 x 4 x 64), doubles
    imgs = 4 * 38 * 8 * 64 + 2 * (4 * 128 * 8 * 64) + 4 * 128 * 4 * 64
    assert L.nocf_workspace_bytes_f64(150, 512, 2) == ((150 + 1) * 512 + 512 * 512 + imgs) * 8
    assert L.nocf_workspace_bytes_f64(4, 32, 1) == 0
    phi, prob = _lib.NocfPhi64(), _lib.NocfProb64()
    alph = (C.c_double * 6)(*[1.0] * 6)
    rc = L.nocf_rollout_f64(C.byref(phi), C.byref(prob), None, 4, 0.0, 1.0, 2, _lib.NOCF_RK4, alph, None, None, None, None, None, None, 0, None)
    assert rc == -1                                          # NOCF_E_NULL: no tensors in the structs
    assert L.nocf_phi_f64(C.byref(phi), None, 4, None, None, None, 0, None) == -1
    assert L.nocf_prob_eval_f64(C.byref(prob), 4, None, None, 4, None, None, None, None) == -1
    assert L.nocf_colsum_f32(None, 10, 4, None, 0, None, 0, None) == -1


def test_product_path_refuses_cpu_tensors():
    import neuraloc_amd as na
    net = na.Phi(2, 8, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.getGrad(torch.zeros(2, 5))
    prob = na.Cross2D(torch.zeros(4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        na.OCflow(torch.zeros(2, 4), net, prob, [0.0, 1.0], 4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "neuraloc_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


def test_state_dict_contract():
    import neuraloc_amd as na
    net = na.Phi(3, 8, 4)
    assert list(net.state_dict().keys()) == ["A", "c.weight", "c.bias", "w.weight", "N.layers.0.weight",
                                             "N.layers.0.bias", "N.layers.1.weight", "N.layers.1.bias",
                                             "N.layers.2.weight", "N.layers.2.bias"]
    assert net.A.shape == (5, 5) and float(net.w.weight.min()) == 1.0 and float(net.c.weight.abs().max()) == 0.0
    with pytest.raises(ValueError):
        na.Phi(1, 8, 4)


def test_checkpoint_layout_roundtrip(tmp_path):
    """{'args': Namespace, 'state_dict'} files, same keys as the reference writes"""
    import argparse
    import neuraloc_amd as na
    from neuraloc_amd.checkpoint import save_checkpoint
    net = na.Phi(2, 16, 4, alph=[300.0, 1e6, 1e5, 1.0, 1.0, 3.0])
    path = tmp_path / "swap2_checkpt.pth"
    save_checkpoint(str(path), net, argparse.Namespace(data="swap2", m=16, nTh=2, alph=net.alph, n_train=8, var0=1.0))
    ck = torch.load(str(path), map_location="cpu", weights_only=False)
    assert set(ck) == {"args", "state_dict"} and ck["args"].data == "swap2"
    assert list(ck["state_dict"]) == list(net.state_dict())
    net2 = na.Phi(2, 16, 4)
    net2.load_state_dict(ck["state_dict"])
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))
