"""pytest configuration: markers and shared helpers."""

from __future__ import annotations

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
for p in (str(ROOT), str(ROOT / "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):  # noqa: ANN001, ANN201
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str) -> dict:
    from safetensors.torch import load_file

    return load_file(str(GOLDEN / name))


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    return GOLDEN
