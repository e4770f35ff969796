"""pytest config: registers the `gpu` marker; GPU tests are skipped when no GPU is visible."""
import os
import sys
from pathlib import Path

os.environ.setdefault("KA_CHECK_ARGS", "1")     # every tensor handed to the C ABI: on the current GPU, contiguous (keisei_amd/_lib.py)

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """Lazy view over one tests/golden/*.npz fixture returning torch tensors."""

    def __init__(self, name: str):
        self._z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)

    def __contains__(self, k):
        return k in self._z.files

    def keys(self):
        return list(self._z.files)

    def np(self, k):
        return self._z[k]

    def __getitem__(self, k) -> torch.Tensor:
        return torch.from_numpy(np.array(self._z[k]))

    def sub(self, prefix: str) -> dict:
        return {k[len(prefix):]: self[k] for k in self._z.files if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return load


@pytest.fixture
def ka_env(monkeypatch):
    """Set / unset a KA_* switch of the library inside one test: the library caches its switches (ka_options_reload)."""
    from keisei_amd import _lib

    class _Env:
        def set(self, name, value):
            monkeypatch.setenv(name, str(value))
            _lib.reload_options()

        def unset(self, name):
            monkeypatch.delenv(name, raising=False)
            _lib.reload_options()

    yield _Env()
    monkeypatch.undo()
    _lib.reload_options()
