import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# the library caches its NOCF_* knobs; the tests switch kernels with them between calls: the Python layer then drops the cache on a change
os.environ["NOCF_ENV_WATCH"] = "1"
# unlisted shapes: no background compilations from the test-suite (the JIT tests switch it on themselves)
os.environ.setdefault("NOCF_JIT", "0")
# the synchronous check of a process's first split-role launches (neuraloc_amd._lib.duo_guard) would turn the forced-timeout tests into
# fallbacks: off in the suite, tested in a child process (tests/test_duo_gpu.py)
os.environ.setdefault("NOCF_DUO_PROBATION", "0")

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
PRETRAINED = ["swap2", "softcorridor", "swap12", "swarm50", "singlequad"]
SYNTHETIC = ["synth_nTh3_midcross4", "synth_nTh4_softcorridor", "synth_nTh3_swarm"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """one tests/golden/<name>.npz: reference inputs + reference outputs."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.meta = json.loads(str(self.z["meta"]))

    def __getitem__(self, k):
        return self.z[k]

    def t(self, k, **kw):
        return torch.from_numpy(np.array(self.z[k])).to(**kw)

    def state_dict(self):
        return {k[3:]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith("sd/")}

    def has(self, k):
        return k in self.z.files


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(params=PRETRAINED + SYNTHETIC)
def golden(request):
    return load_golden(request.param)


@pytest.fixture(params=PRETRAINED)
def golden_pretrained(request):
    return load_golden(request.param)


@pytest.fixture(autouse=True)
def _nan_filled_free_blocks(request):
    """NOCF_TEST_POISON=1 (a debugging run of the GPU suite, not the default -- it adds ~1 s per test): the caching allocator's free blocks are
    filled with NaN in front of every GPU test, so a kernel that reads a row nobody wrote (or reads it before its producer) computes with NaN
    instead of with an earlier test's values.  The tape / adjoint tests do this themselves, always (tests/util_hip.poison_allocator)."""
    if os.environ.get("NOCF_TEST_POISON", "0") == "1" and request.node.get_closest_marker("gpu") is not None and torch.cuda.is_available():
        from util_hip import poison_allocator
        poison_allocator(torch.device("cuda:0"), big=2)
    yield
