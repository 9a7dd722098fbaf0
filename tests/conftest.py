"""Shared test plumbing.

* registers the ``gpu`` marker (tests that need a real MI355X);
* loads the product binding ``lbm-asynchronous_amd/__init__.py`` (hyphenated directory, so via
  importlib) as ``lbm_asynchronous_amd``;
* builds the CPU oracle (oracle/Makefile) on demand and exposes it through
  ``tests/oracle_binding.py``.  Only tests may touch ``oracle/``.
"""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "lbm-asynchronous_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    if "lbm_asynchronous_amd" in sys.modules:
        return sys.modules["lbm_asynchronous_amd"]
    spec = importlib.util.spec_from_file_location(
        "lbm_asynchronous_amd", os.path.join(PKG_DIR, "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["lbm_asynchronous_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def lbm():
    mod = load_package()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.load()


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


def dataset(name):
    """(Params-tuple, obstacle map) of one of the reference's four data sets (tests/golden/inputs)."""
    mod = load_package()
    p = mod.read_params(os.path.join(GOLDEN, "inputs", f"input_{name}.params"))
    ob = mod.read_obstacles(os.path.join(GOLDEN, "inputs", f"obstacles_{name}.dat"), p.nx, p.ny)
    return p, ob


@pytest.fixture(scope="session")
def datasets():
    return dataset
