"""pytest plugin of the conformance harness (dev container only -- needs /root/reference; nothing here ships or
travels to the GPU box).

It makes ``import keisei`` resolve to a synthetic package whose hot-path modules ARE this build's modules

    keisei.training.{model_registry, value_adapter, algorithm_registry, gae, katago_ppo, distributed}
    keisei.training.models.{katago_base, se_resnet, base, mlp, resnet, transformer}
    keisei.sl.{dataset, trainer}

while every other submodule (config, db, checkpoint, katago_loop, opponent_store, ...) is found in the reference tree
through ``__path__``.  The reference's own test files and its unmodified training loop then run *on top of the build*
(SURVEY 8c "Conformance harness").  Also closes the two py3.10 stdlib gaps of the reference (tomllib, enum.StrEnum).
"""
from __future__ import annotations

import enum
import importlib
import os
import sys
import types
from pathlib import Path

REF = Path(os.environ.get("KEISEI_REFERENCE", "/root/reference"))
REPO = Path(__file__).resolve().parents[2]
sys.dont_write_bytecode = True            # the reference tree is read-only

HOT = {
    "keisei.training.model_registry": "keisei_amd.training.model_registry",
    "keisei.training.value_adapter": "keisei_amd.training.value_adapter",
    "keisei.training.algorithm_registry": "keisei_amd.training.algorithm_registry",
    "keisei.training.gae": "keisei_amd.training.gae",
    "keisei.training.katago_ppo": "keisei_amd.training.katago_ppo",
    "keisei.training.distributed": "keisei_amd.training.distributed",
    "keisei.training.models.katago_base": "keisei_amd.training.models.katago_base",
    "keisei.training.models.se_resnet": "keisei_amd.training.models.se_resnet",
    "keisei.training.models.base": "keisei_amd.training.models.base",
    "keisei.training.models.mlp": "keisei_amd.training.models.mlp",
    "keisei.training.models.resnet": "keisei_amd.training.models.resnet",
    "keisei.training.models.transformer": "keisei_amd.training.models.transformer",
    "keisei.sl.dataset": "keisei_amd.sl.dataset",
    "keisei.sl.trainer": "keisei_amd.sl.trainer",
}


def _package(name: str, path: Path) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__path__ = [str(path)]            # non-hot submodules are found in the reference tree
    mod.__package__ = name
    sys.modules[name] = mod
    return mod


def install() -> None:
    if "keisei" in sys.modules and getattr(sys.modules["keisei"], "_conformance_shim", False):
        return
    if not (REF / "keisei").is_dir():
        raise RuntimeError(f"conformance harness: {REF}/keisei not found (dev container only)")
    if str(REPO) not in sys.path:
        sys.path.insert(0, str(REPO))
    # py3.10: tomllib / StrEnum
    if "tomllib" not in sys.modules:
        try:
            import tomllib  # noqa: F401
        except ModuleNotFoundError:
            sys.modules["tomllib"] = importlib.import_module("tomli")
    if not hasattr(enum, "StrEnum"):
        class StrEnum(str, enum.Enum):
            def __str__(self) -> str:
                return str(self.value)

            @staticmethod
            def _generate_next_value_(name, start, count, last_values):
                return name.lower()

        enum.StrEnum = StrEnum
    root = _package("keisei", REF / "keisei")
    root._conformance_shim = True
    root.__version__ = "conformance-shim"
    training = _package("keisei.training", REF / "keisei" / "training")
    models = _package("keisei.training.models", REF / "keisei" / "training" / "models")
    sl = _package("keisei.sl", REF / "keisei" / "sl")
    root.training, root.sl, training.models = training, sl, models
    for alias, real in HOT.items():
        mod = importlib.import_module(real)
        sys.modules[alias] = mod
        parent, _, leaf = alias.rpartition(".")
        setattr(sys.modules[parent], leaf, mod)


install()


def _graft_loop_helpers() -> None:
    """KEISEI_CONFORMANCE_LOOP=1: the rollout helpers of the reference's katago_loop.py (split_merge_step,
    PendingTransitions, the perspective corrections ...) are replaced by this build's (keisei_amd.training.katago_loop)
    INSIDE the reference's module, so that the reference's tests of those helpers -- and its loop -- run on them."""
    ref_loop = importlib.import_module("keisei.training.katago_loop")
    mine = importlib.import_module("keisei_amd.training.katago_loop")
    for name in mine.__all__:
        setattr(ref_loop, name, getattr(mine, name))


if os.environ.get("KEISEI_CONFORMANCE_LOOP") == "1":
    _graft_loop_helpers()


def pytest_report_header(config):
    return [f"keisei conformance shim: hot-path modules -> keisei_amd ({len(HOT)} aliases), the rest from {REF}"]
