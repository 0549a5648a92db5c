import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_resource(name):
    with open(os.path.join(GOLDEN, "resources", name), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def fixture_sources():
    return {k: read_resource(k) for k in ["lex.csv", "matrix.def", "char.def", "unk.def", "user.csv"]}


@pytest.fixture(scope="session")
def tokenize_golden():
    with open(os.path.join(GOLDEN, "tokenize_golden.json"), encoding="utf-8") as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def unit_golden():
    with open(os.path.join(GOLDEN, "unit_golden.json"), encoding="utf-8") as f:
        return json.load(f)
