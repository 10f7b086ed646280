import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    return entry.load_oracle()


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(GOLDEN, "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def svc(pkg):
    """One HIP service for the whole GPU session.  Fails loudly (no fallback) if the library is missing."""
    s = pkg.HipCompressionService(chunk_size_mb=1, device=0)
    yield s
    s.close()


def recipe_bytes(orc, recipe):
    """Inputs of tests/golden/reference_vectors.json, by recipe name."""
    if recipe == "AAAABBBBCCCCDDDD":
        return np.frombuffer(b"AAAABBBBCCCCDDDD", dtype=np.uint8)
    if recipe == "A*1024":
        return np.full(1024, 0x41, dtype=np.uint8)
    if recipe == "Hello World! *100":
        return np.frombuffer(("Hello World! " * 100).encode(), dtype=np.uint8)
    if recipe == "Test data for integrity check":
        return np.frombuffer(recipe.encode(), dtype=np.uint8)
    if recipe.startswith("java.util.Random(42).nextBytes("):
        return orc.java_random_bytes(42, int(recipe.split("(")[-1].rstrip(")")))
    if recipe.startswith("i%256"):
        return (np.arange(3 * 1024 * 1024) % 256).astype(np.uint8)
    if recipe.startswith("'A'+(i/100)%26"):
        return (0x41 + (np.arange(512 * 1024) // 100) % 26).astype(np.uint8)
    if recipe.startswith("A*2097152"):
        return np.full(2 * 1024 * 1024, 0x41, dtype=np.uint8)
    raise KeyError(recipe)


def pin_input(orc, pin):
    if "file" in pin:
        return np.fromfile(os.path.join(GOLDEN, pin["file"]), dtype=np.uint8)
    return recipe_bytes(orc, pin["recipe"])
