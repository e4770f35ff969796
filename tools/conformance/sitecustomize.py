"""Child processes of the conformance harness (mp.spawn ranks of the reference's DDP integration test) start a fresh
interpreter: with tools/conformance on PYTHONPATH and KEISEI_CONFORMANCE=1 they install the same module aliases."""
import os

if os.environ.get("KEISEI_CONFORMANCE") == "1":
    import keisei_shim_plugin  # noqa: F401
