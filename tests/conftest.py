import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("racing-slam_amd")


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def rs(pkg):
    return pkg.rsgpu


@pytest.fixture(scope="session")
def ctx(rs):
    """GPU context; only gpu-marked tests may request it."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test selected without a GPU: the product path has no CPU fallback")
    c = rs.Context(0)
    yield c
    c.close()


def to_np(t):
    return t.detach().cpu().numpy()
