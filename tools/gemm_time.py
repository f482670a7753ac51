#!/usr/bin/env python3
"""Time of ONE f16x3 GEMM shape (ms, mean of 10 after 3 warm-ups): tools/gemm_time.py M K N [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import ops_gemm  # noqa: E402

M, K, N = (int(v) for v in sys.argv[1:4])
hb = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if hb:
    a = torch.randn(M, hb, K, device="cuda").permute(1, 0, 2)
    w = torch.randn(hb, K, N, device="cuda") * 0.05
else:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(K, N, device="cuda") * 0.05
am = ops_gemm.amax(a)
ws = ops_gemm.presplit(w)
for _ in range(3):
    ops_gemm._forward(a, w, None, None, 0, 0.0, am, ws)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    ops_gemm._forward(a, w, None, None, 0, 0.0, am, ws)
e.record()
torch.cuda.synchronize()
print(f"[{M},{K}]x[{K},{N}]" + (f" x{hb} heads" if hb else "") + f": {s.elapsed_time(e) / 10:.3f} ms")
