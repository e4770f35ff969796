"""Runs the default conv3x3 configuration a few times (for rocprofv3 --pmc runs)."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
B, C = 4096, 256
x = torch.randn(B, 81, C, device='cuda').bfloat16(); w = torch.randn(C, C, 3, 3, device='cuda') / 48
wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device='cuda')
_lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, _lib.stream_ptr())
out = torch.empty_like(x); rows = _lib.query("ka_conv3x3_sqpart_rows", B)
bsum = torch.empty(B, C, device='cuda'); sq = torch.empty(rows, C, device='cuda')
for _ in range(5):
    _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, 1, _lib.stream_ptr())
torch.cuda.synchronize()
