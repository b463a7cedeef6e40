import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "chinesechess-alphazero_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """libczero.so is a build artefact (git-ignored): compile it when it is missing or stale and hipcc is here
    (cross-compiles for gfx950 without a GPU).  The oracle builds itself on first use."""
    import importlib.util
    import shutil
    spec = importlib.util.spec_from_file_location("czero_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.needs_build() and shutil.which("hipcc"):
        mod.build(verbose=False)


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def positions_1k():
    return load_golden("positions_1k.json")["positions"]


@pytest.fixture(scope="session")
def catch_cases():
    return load_golden("catch_cases.json")["cases"]


@pytest.fixture(scope="session")
def known_answers():
    return load_golden("known_answers.json")
