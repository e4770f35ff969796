"""ctypes binding of libkeisei_amd.so (the C ABI declared in include/keisei_amd.h).

The library is the only compute backend for CUDA/HIP tensors: if it is missing or fails to
load, every GPU entry point raises -- there is no eager/PyTorch fallback on the GPU path.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path
from typing import Optional

import torch

# KEISEI_AMD_LIB: alternative build of the same library (A/B timing of kernel changes); never a fallback
_LIB_PATH = Path(os.environ.get("KEISEI_AMD_LIB") or Path(__file__).resolve().parent / "libkeisei_amd.so")

# signature strings: p = pointer (torch tensor | int | None), i = int, f = float, d = double, q = long long
_SIGS = {
    "ka_conv3x3_fwd": "pppppp i pp iii i p",
    "ka_conv3x3_fwd_keep_supported": "iiii",
    "ka_conv3x3_fwd_keep": "pppppp i pp p iii i p",
    "ka_conv3x3_dgrad_fused": "ppppp pp ppppp pp iii i p",
    "ka_conv3x3_dgrad_gated_supported": "iiiii",
    "ka_conv3x3_dgrad_fused_gated": "pppppp pp ppppp pp iii i p",
    "ka_conv3x3_sqpart_rows": "i",
    "ka_debug_conv_stamps": "p",
    "ka_pack_conv3x3": "pp iiii i i p",
    "ka_pack_conv3x3_multi": "p i q i p",
    "ka_wgrad_splits": "iiii",
    "ka_conv3x3_wgrad": "ppppp i pp iiii i i i p",
    "ka_obs_to_nhwc": "ppp iii i p",
    "ka_nhwc_to_nchw": "pp ii i p",
    "ka_bn_reduce": "p i p i i pp p",
    "ka_reduce_workspace_doubles": "i",
    "ka_pair_reduce": "pp ii pp p",
    "ka_sync_reduce": "p i p i i d ppp p",
    "ka_bn_coeffs": "p d p pppp p ff pppp i p",
    "ka_bn_coeffs_parts": "p d pppp p ff pppp i p",
    "ka_bn_bwd_coeffs_parts": "p d ppp ppp i i p",
    "ka_bn_reduce_coeffs": "p i p i i pp d pppp p ff pppp p",
    "ka_pair_reduce_bwd_coeffs": "pp i i pp d ppp ppp i p",
    "ka_bn_eval_coeffs": "pppp f pp i p",
    "ka_bn_bwd_coeffs": "pp d p ppp ppp i i p",
    "ka_affine_rows": "ppp f p ii p",
    "ka_bn_bwd_apply": "pppp ii i p",
    "ka_block_tail_fwd": "ppppp pp ii i p",
    "ka_block_tail_fwd_se_supported": "iii",
    "ka_block_tail_fwd_se": "pppppppp ppp ppp iii i p",
    "ka_pool_fwd": "pp ii i p",
    "ka_tail_bwd_reduce": "pppppp p ii i p",
    "ka_tail_bwd_dz": "ppppppp ppp ii i p",
    "ka_tail_bwd_fused_supported": "iii",
    "ka_tail_bwd_fused": "ppppp pppppp p pppp iii i p",
    "ka_relu_bn_bwd_reduce": "pppppp ppp ii i p",
    "ka_block_dx": "pppppp p ii i p",
    "ka_block_dx_tail_bwd_supported": "iii",
    "ka_block_dx_tail_bwd": "pppppp p ppppppppp p pppp iii i p",
    "ka_block_dx_tail_bwd_du": "ppppp p ppppppppp p pppp iii i p",
    "ka_block_dx_tail_bwd_du_gate": "ppppp p ppppppppp pp pppp iii i p",
    "ka_gemm": "pppp iii iii ii iii i i i p",
    "ka_reduce_slabs": "pp i q i p",
    "ka_reduce_slabs2": "pp q pp q i p",
    "ka_colsum": "pppp ii i p",
    "ka_relu_mask": "pp q p",
    "ka_rows_affine_relu": "pppp q i p",
    "ka_rows_bn_bwd": "pppp q i i p",
    "ka_rows_bn_sums": "pppp pp ii i p",
    "ka_rows_sq_sums": "ppp ii i p",
    "ka_policy_loss": "pppppp pppp pp fff iii p",
    "ka_masked_softmax": "ppppp iii p",
    "ka_policy_sample": "pi pi q pp f ppp pp ii p",
    "ka_policy_ce": "ppp ppp p f ii p",
    "ka_value_loss": "ppppp pp pp pp p ffff i i p",
    "ka_scalar_value": "pp f p i p",
    "ka_adam_chunk": "",
    "ka_clip_adam_step": "ppp i ppp ppp fffff p",
    "ka_gae": "ppppp pp ii dd i p",
    "ka_normalize_advantages": "pp q p",
    "ka_bn_eval_coeffs_multi": "p ii p",
    "ka_gemm_grouped_wgrad": "p ii p",
    "ka_fc_chain_supported": "iiii",
    "ka_fc_chain": "ppp f pppp ppp iiiii p",
    "ka_fc_chain_bwd": "pppppp iiii p",
    "ka_transpose_multi": "p ii p",
    "ka_mask_words": "i",
    "ka_rollout_append": "pppppppppppp pppppppppppp p iii p",
    "ka_rollout_append_packed": "pppppppppppp pppppppppppp p iii p",
    "ka_unpack_mask_bits": "ppp ii p",
    "ka_pack_mask_bits": "pp ii p",
    "ka_pending_open": "pppppppp ppppppppp p iii p",
    "ka_pending_accumulate": "ppp i p",
    "ka_pending_settle": "ppppppppp ppp i p ppppppppppp p iii p",
    "ka_tower_eval_supported": "iiii",
    "ka_tower_eval": "ppppp iiiii i p",
    "ka_shogi_env_state_bytes": "",
    "ka_shogi_env_action_space": "i",
    "ka_shogi_env_reset": "ppp ii ii pppp i p",
    "ka_shogi_env_step": "pppp ii ii ppp ppp ppp pp ppp pp p",
    "ka_tf_gemm_nt": "ppppp iii iii iii f q p",
    "ka_tf_gemm_nt_slabs": "ii",
    "ka_tf_gemm_nt_masked": "pppp iii iii f q p",
    "ka_tf_gemm_tn": "ppp iii iii i p",
    "ka_tf_gemm_tn_bias": "pppp iii iii i p",
    "ka_tf_gemm_tn_slabs": "ii",
    "ka_tf_transpose_pad": "pp iiii i p",
    "ka_tf_cast_pad": "pp q iii i p",
    "ka_tf_weights16_multi": "p ii p",
    "ka_tf_add_pos": "ppp ii i p",
    "ka_tf_pos_grad": "pppp ii i p",
    "ka_tf_layernorm_fwd": "pppppp q i f i p",
    "ka_tf_layernorm_parts": "q",
    "ka_tf_layernorm_bwd": "pppppp pppp q i i p",
    "ka_tf_layernorm_bwd_drop": "pppppp pp f q ppp q i i p",
    "ka_tf_drop_apply": "pppp q f q i p",
    "ka_tf_colsum": "ppp q ii i p",
    "ka_tf_mean_pool": "pp ii i p",
    "ka_tf_head_grad": "ppp ii i p",
    "ka_tf_tanh": "p q p",
    "ka_tf_tanh_bwd": "ppp q p",
    "ka_tf_attention_fwd": "ppp iii f q i p",
    "ka_tf_attention_bwd": "pppp iii f q i p",
    "ka_tf_attention_bwd_o": "ppppp iii f q i p",
    "ka_version": "",
    "ka_options_reload": "",
}
_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "f": ctypes.c_float, "d": ctypes.c_double, "q": ctypes.c_longlong}   # q also carries 64-bit seeds

_lib: Optional[ctypes.CDLL] = None
_load_error: Optional[str] = None


class KeiseiHipError(RuntimeError):
    """Raised when a libkeisei_amd.so entry point reports failure (or the library is absent)."""


def _load() -> ctypes.CDLL:
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise KeiseiHipError(_load_error)
    if not _LIB_PATH.exists():
        _load_error = (f"{_LIB_PATH} is missing: build it with `python -m keisei_amd.build` "
                       "(hipcc --offload-arch=gfx950). There is no fallback for GPU tensors.")
        raise KeiseiHipError(_load_error)
    try:
        lib = ctypes.CDLL(str(_LIB_PATH))
    except OSError as e:  # pragma: no cover
        _load_error = f"cannot load {_LIB_PATH}: {e}"
        raise KeiseiHipError(_load_error) from e
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = [_CT[c] for c in sig.replace(" ", "")]
        fn.restype = ctypes.c_int
    lib.ka_last_error.restype = ctypes.c_char_p
    lib.ka_last_error.argtypes = []
    lib.ka_target_arch.restype = ctypes.c_char_p
    lib.ka_target_arch.argtypes = []
    _lib = lib
    return lib


def reload_options() -> int:
    """The library reads its KA_* switches from the environment once (first launch).  A process that changes one afterwards
    -- tests, A/B tools flipping a kernel form in place -- calls this to have them read again; returns how many are set."""
    _QUERIES.clear()
    return int(_load().ka_options_reload())


def library_path() -> Path:
    return _LIB_PATH


def exported_symbols() -> list:
    """Names this binding expects the shared library to export (checked by the CPU tests)."""
    return sorted(list(_SIGS) + ["ka_last_error", "ka_target_arch"])


def available() -> bool:
    try:
        _load()
        return True
    except KeiseiHipError:
        return False


# KA_CHECK_ARGS=1 (the GPU test-suite sets it): every tensor handed to the C ABI must live on a GPU and be contiguous,
# and all tensors of one call must share one device, which must be the CURRENT device (launches go to the current
# device's stream).  The C ABI itself sees only raw pointers, so this is the last place where a host tensor, a strided
# view or a tensor of another card can be caught instead of being misread.  Off by default: ~1000 launches per step.
_CHECK = os.environ.get("KA_CHECK_ARGS", "0") == "1"


def _check_tensors(name: str, args) -> None:
    dev = None
    for i, a in enumerate(args):
        if not isinstance(a, torch.Tensor):
            continue
        if not a.is_cuda:
            raise KeiseiHipError(f"{name}: argument {i} is a {a.device} tensor (the C ABI takes device pointers)")
        if not a.is_contiguous():
            raise KeiseiHipError(f"{name}: argument {i} is not contiguous (shape {tuple(a.shape)}, strides {a.stride()})")
        if dev is None:
            dev = a.device
        elif a.device != dev:
            raise KeiseiHipError(f"{name}: tensors on different devices ({dev} and {a.device})")
    if dev is not None and dev.index != torch.cuda.current_device():
        raise KeiseiHipError(f"{name}: tensors live on {dev} but the current device is cuda:{torch.cuda.current_device()} "
                             "(wrap the call in torch.cuda.device(tensor.device))")


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.data_ptr()
    return int(x)


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


_PLANS: dict = {}      # name -> (bound ctypes function, indices of the pointer arguments, argument count)
_Tensor = torch.Tensor
_cur_dev = torch.cuda.current_device


def _plan(name: str):
    lib = _load()
    sig = _SIGS[name].replace(" ", "")
    ent = (getattr(lib, name), tuple(i for i, c in enumerate(sig) if c == "p"), len(sig))
    _PLANS[name] = ent
    return ent


def call(name: str, *args) -> None:
    """Call a status-returning entry point; raises KeiseiHipError with ka_last_error() on failure.
    (A step is ~1000 of these and small configurations are bound by the host's launch rate: the per-name work -- the cleaned
    signature, the bound function -- is done once, the per-call work is one pass over the pointer arguments.)"""
    ent = _PLANS.get(name)
    if ent is None:
        ent = _plan(name)
    fn, ptr_idx, n = ent
    if len(args) != n:
        raise TypeError(f"{name}: expected {n} arguments, got {len(args)}")
    conv = list(args)
    dev = None
    for i in ptr_idx:
        a = conv[i]
        if a is None:
            continue
        if type(a) is _Tensor or isinstance(a, _Tensor):
            if dev is None and a.is_cuda:
                dev = a.device.index
            conv[i] = a.data_ptr()
        else:
            conv[i] = int(a)
    # launches go to the CURRENT device (and set per-device kernel attributes there): a model that lives on another
    # card than the current one (the reference places league opponents on a second GPU, katago_loop.py:371-422) is
    # launched under that card's device guard
    if dev is not None and dev != _cur_dev():
        with torch.cuda.device(dev):
            if _CHECK:
                _check_tensors(name, args)
            rc = fn(*conv)
    else:
        if _CHECK:
            _check_tensors(name, args)
        rc = fn(*conv)
    if rc != 0:
        raise KeiseiHipError(f"{name} failed ({rc}): {_load().ka_last_error().decode()}")


_QUERIES: dict = {}    # (name, args) -> value: the queries are pure functions of their (shape) arguments and of the KA_* switches


def query(name: str, *args) -> int:
    """Call an int-returning pure query (no status convention); results are remembered until reload_options()."""
    key = (name, args)
    v = _QUERIES.get(key)
    if v is None:
        v = _QUERIES[key] = int(getattr(_load(), name)(*args))
    return v


DTYPE_F32, DTYPE_BF16 = 0, 1


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return DTYPE_F32
    if dt == torch.bfloat16:
        return DTYPE_BF16
    raise KeiseiHipError(f"unsupported activation dtype {dt} (float32 or bfloat16)")
