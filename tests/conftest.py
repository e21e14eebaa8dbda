import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


os.environ.setdefault("MSS_LINEAR_STRICT", "1")   # tests: a float32 CUDA Linear outside the MFMA kernels' shapes raises (linear._library_route)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def deeplab_params():
    from multishiftseg_amd import synth
    return synth.deepwv3plus_params(0)


@pytest.fixture(autouse=True)
def _fresh_env_switches():
    """libmss_hip.so caches the MSS_* environment switches per call site (csrc/mss_common.h MSS_ENV_INT); every test starts
    from a re-read, whatever the previous one left behind."""
    from multishiftseg_amd import _lib
    _lib.reset_env_cache()
    yield
    _lib.reset_env_cache()


@pytest.fixture(params=["native", "bf16x3"])
def gemm_route(request):
    """Every reference-fixture / oracle test that exercises a GEMM takes this fixture and runs twice, with the SAME bounds: on the
    native fp32 MFMA kernels and on the split-bf16 route (csrc/gemm_bf16x3.hip: operands as three bf16 terms, six bf16 MFMAs per
    block, fp32 accumulate) -- VERDICT r04 next #1."""
    from multishiftseg_amd import kernels as K
    K.set_gemm_route(request.param)
    yield request.param
    K.set_gemm_route(None)


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch whose setenv / delenv also tell the library to re-read its cached switches."""
    from multishiftseg_amd import _lib
    for name in ("setenv", "delenv"):
        orig = getattr(monkeypatch, name)

        def wrapped(*a, _orig=orig, **k):
            _orig(*a, **k)
            _lib.reset_env_cache()
        setattr(monkeypatch, name, wrapped)
    return monkeypatch
