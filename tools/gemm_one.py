#!/usr/bin/env python3
"""Runs the split GEMM a few times on one DISGAT shape (for rocprofv3 PMC passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import ops_gemm  # noqa: E402

M, K, N = 1_000_000, int(sys.argv[1]), int(sys.argv[2])
a = torch.randn(M, K, device="cuda")
w = torch.randn(K, N, device="cuda") * 0.05
for _ in range(4):
    ops_gemm._forward(a, w, None, None, 0, 0.0)
torch.cuda.synchronize()
