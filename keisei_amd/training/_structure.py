"""A process-wide version counter of nn.Module STRUCTURE: bumped whenever any module registers a submodule, a parameter or a
buffer (torch's global registration hooks), i.e. on `convert_sync_batchnorm`, a replaced head, `module.weight = Parameter(...)`,
a freshly built model.  Host-side caches of flattened module / tensor lists (the rollout path walks ~450 modules per call
otherwise) are valid exactly while the counter stands still; `load_state_dict` copies in place and does not bump it."""
from __future__ import annotations

from torch.nn.modules import module as _m

_version = [0]


def _bump(*_args, **_kw):
    _version[0] += 1
    return None          # keep what is being registered unchanged


_m.register_module_module_registration_hook(_bump)
_m.register_module_parameter_registration_hook(_bump)
_m.register_module_buffer_registration_hook(_bump)


def structure_version() -> int:
    return _version[0]
