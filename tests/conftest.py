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
