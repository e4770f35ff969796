"""Runs the weight-gradient kernel a few times at the headline shape (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
B, C = 4096, 256
x = torch.randn(B, 81, C, device='cuda').bfloat16(); dy = torch.randn(B, 81, C, device='cuda').bfloat16()
ns = _lib.query("ka_wgrad_splits", B, C, C, 0)
slab = torch.empty(ns * 9 * C * C, device='cuda'); dw = torch.empty(C, C, 3, 3, device='cuda')
for _ in range(5):
    _lib.call("ka_conv3x3_wgrad", dy, x, None, None, None, 0, slab, dw, B, C, C, C, 0, 0, 1, _lib.stream_ptr())
torch.cuda.synchronize()
