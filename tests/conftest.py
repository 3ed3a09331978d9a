import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from __graft_entry__ import load_package, load_oracle, build  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mi():
    build()
    return load_package()


@pytest.fixture(scope="session")
def orc(mi):
    o = load_oracle()
    o.lib(mi.abi)
    return o


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible: the engine has no CPU fallback")
    return torch


def small_cbox(mi, res=32, **film):
    d = mi.cornell_box()
    d["sensor"]["film"]["width"] = res
    d["sensor"]["film"]["height"] = res
    d["sensor"]["film"].update(film)
    return d


@pytest.fixture(scope="session")
def cbox32(mi):
    return mi.load_dict(small_cbox(mi, 32))
