import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def bpp():
    """the product package (hyphenated directory name -> importlib)"""
    return importlib.import_module("bulletproofs-plus_amd")


@pytest.fixture(scope="session")
def engine(bpp):
    """one bpp_ctx on device 0; fails loudly if the HIP extension or the GPU is missing"""
    eng = bpp.Engine(0)
    yield eng
    eng.close()


@pytest.fixture
def opt(engine):
    """set per-context knobs of the session engine for one test (bpp_ctx_set_option); restored afterwards"""
    touched = []

    def set_option(name, value):
        engine.set_option(name, value)
        touched.append(name)
    yield set_option
    for name in touched:
        engine.set_option(name, -1)
