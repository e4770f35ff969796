"""Stand-alone GlobalPoolBiasBlock on the GPU (NCHW fp32 in/out at the API boundary).

Used when a caller invokes ``model.blocks[i](x)`` directly, as the reference's
scripts/profile_hotpath.py:630 does; inside SEResNetModel the blocks are driven by
keisei_amd/hip/seresnet.py without layout conversions.
"""
from __future__ import annotations

import torch
from torch import nn

from keisei_amd import _lib
from keisei_amd.hip.seresnet import SEResNetEngine, _Saved

_call = _lib.call


class _Shim(nn.Module):
    """Presents one block as a 1-block tower to the engine's helpers."""

    def __init__(self, block):
        super().__init__()
        object.__setattr__(self, "blocks", [block])


def run_block(block: nn.Module, x: torch.Tensor) -> torch.Tensor:
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in block.parameters())):
        raise _lib.KeiseiHipError(
            "stand-alone GlobalPoolBiasBlock on the GPU is inference-only; differentiate through SEResNetModel "
            "(single fused autograd node) instead")
    B, C = x.shape[0], x.shape[1]
    dev = x.device
    st = _lib.stream_ptr(dev)
    T = torch.bfloat16 if (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16) \
        else torch.float32
    code = _lib.dtype_code(T)
    eng = getattr(block, "_hip_engine", None)
    if eng is None:
        eng = SEResNetEngine(_Shim(block))
        eng._conv_layers = lambda: iter((("blocks.0.conv1", block.conv1), ("blocks.0.conv2", block.conv2)))
        object.__setattr__(block, "_hip_engine", eng)
    packs = eng._get_packs(T, dev)
    train = block.training
    xin = torch.empty(B, 81, C, dtype=T, device=dev)
    _call("ka_obs_to_nhwc", x.float().contiguous(), None, xin, B, C, C, code, st)
    pool = torch.empty(B, 4 * C, device=dev)
    _call("ka_pool_fwd", xin, pool, B, C, code, st)
    rows = _lib.query("ka_conv3x3_sqpart_rows", B)
    y1 = torch.empty_like(xin); bs = torch.empty(B, C, device=dev); sq = torch.empty(rows, C, device=dev)
    _call("ka_conv3x3_fwd", xin, packs["blocks.0.conv1"][0], y1, None, None, None, 0, bs if train else None,
          sq if train else None, B, C, C, code, st)
    sc1, sh1, _, _ = eng._bn_forward(block.bn1, bs, B, sq, rows, C, B * 81, train, dev, st)
    g = eng._linear(eng._linear(pool, block.global_fc[0], 1, st), block.global_fc[2], 0, st)
    y2 = torch.empty_like(xin); bs2 = torch.empty(B, C, device=dev)
    _call("ka_conv3x3_fwd", y1, packs["blocks.0.conv2"][0], y2, sc1, sh1, g, 1, bs2, sq if train else None, B, C, C, code, st)
    sc2, sh2, _, _ = eng._bn_forward(block.bn2, bs2, B, sq, rows, C, B * 81, train, dev, st)
    sqz = torch.empty(B, C, device=dev)
    _call("ka_affine_rows", bs2, sc2, sh2, 1.0 / 81.0, sqz, B, C, st)
    se = eng._linear(eng._linear(sqz, block.se_fc1, 1, st), block.se_fc2, 0, st)
    out = torch.empty_like(xin)
    _call("ka_block_tail_fwd", y2, sc2, sh2, se, xin, out, None, B, C, code, st)
    res = torch.empty(B, C, 9, 9, device=dev)
    _call("ka_nhwc_to_nchw", out, res, B, C, code, st)
    return res
