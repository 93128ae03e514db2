"""pytest configuration: the `gpu` marker (tests that need an MI355X) and shared fixtures."""
from __future__ import annotations

import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / "tests" / "golden"
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """Path of the in-tree C-ABI library, (re)built with hipcc if stale (cross-compiles without a GPU)."""
    from chimeralm_amd import build

    return build.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
