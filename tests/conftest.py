import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
STAN = os.path.join(ROOT, "klhr_amd", "stan")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_generate_tests(metafunc):
    """Every GPU test runs three times: with the library's own choice of hot-path kernels ("auto"), with the monolithic
    kernels forced (KLHR_KERNEL=mono: lane-per-chain for D <= 26, octet-per-chain above) and with the split-phase pipeline
    forced wherever it is supported (KLHR_KERNEL=split: funnel / ill-normal / normal, every variant but the slice
    sampler).  Tests that pick a mapping themselves (test_gpu_matrix.py) override the variable.  CPU tests run once."""
    if "kernel_mode" in metafunc.fixturenames:
        gpu = metafunc.definition.get_closest_marker("gpu") is not None
        metafunc.parametrize("kernel_mode", ["auto", "mono", "split"] if gpu else ["auto"], indirect=True)


@pytest.fixture(autouse=True)
def kernel_mode(request, monkeypatch):
    mode = getattr(request, "param", "auto")
    if mode == "auto":
        monkeypatch.delenv("KLHR_KERNEL", raising=False)
    else:
        monkeypatch.setenv("KLHR_KERNEL", mode)
    yield mode


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def model_spec(name):
    """fixture name -> (stan stem, data dict)"""
    if name == "funnel10":
        return "funnel", {"D": 9}
    if name == "funnel100":
        return "funnel", {"D": 99}
    if name == "illnormal100":
        return "ill-normal", {"D": 100}
    if name == "normal2":
        return "normal", {"D": 2}
    if name == "normal100":
        return "normal", {"D": 100}
    if name == "ar1x100":
        return "ar1", {"N": 100}
    if name == "earnings":
        with open(os.path.join(STAN, "earnings.json")) as f:
            return "earnings", json.load(f)
    if name == "glmm45":
        with open(os.path.join(STAN, "glmm-poisson.json")) as f:
            return "glmm-poisson", json.load(f)
    raise KeyError(name)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def has_gpu():
    return os.path.exists("/dev/kfd")
