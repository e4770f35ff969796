"""A process-wide version counter of nn.Module STRUCTURE: bumped whenever any module registers a submodule, a parameter or a
buffer (torch's global registration hooks), i.e. on `convert_sync_batchnorm`, a replaced head, `module.weight = Parameter(...)`,
a freshly built model -- and whenever one is REMOVED: `del m.sub`, `m.buf = None` on a registered buffer or parameter,
`m._modules.pop(...)` / `m._buffers.pop(...)` (the dictionaries themselves are watched).  Host-side caches of flattened module /
tensor lists (the rollout path walks ~450 modules per call otherwise) are valid exactly while the counter stands still;
`load_state_dict` copies in place and does not bump it.  A removed module's storage can be freed and reused, so a cache keyed on
stale tensors could replay a captured graph against dead pointers: removals count."""
from __future__ import annotations

from torch.nn.modules import module as _m

_version = [0]


def _bump(*_args, **_kw):
    _version[0] += 1
    return None          # keep what is being registered unchanged


_m.register_module_module_registration_hook(_bump)
_m.register_module_parameter_registration_hook(_bump)
_m.register_module_buffer_registration_hook(_bump)


# removals: nn.Module.__delattr__ (del m.sub / del m.weight / del m.buf) and assignments of None to a registered name
# (nn.Module.__setattr__ stores None straight into _parameters / _buffers / _modules without calling a registration hook)
_orig_delattr = _m.Module.__delattr__
_orig_setattr = _m.Module.__setattr__


def _delattr(self, name):
    _version[0] += 1
    return _orig_delattr(self, name)


def _setattr(self, name, value):
    if value is None:
        d = self.__dict__
        if any(name in d.get(k, ()) for k in ("_parameters", "_buffers", "_modules")):
            _version[0] += 1
    return _orig_setattr(self, name, value)


_m.Module.__delattr__ = _delattr
_m.Module.__setattr__ = _setattr


def structure_fingerprint(model) -> tuple:
    """(#modules, #parameters, #buffers) of a tree: a cheap backstop for edits that go around every hook
    (`m._modules.pop(name)`); callers compare it every few dozen calls."""
    return (sum(1 for _ in model.modules()), sum(1 for _ in model.parameters()), sum(1 for _ in model.buffers()))


def structure_version() -> int:
    return _version[0]
