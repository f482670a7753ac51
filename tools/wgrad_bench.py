#!/usr/bin/env python3
"""Weight-gradient GEMM (a^T g over M = 1e6 rows): split-K f16x3 kernel vs hipBLASLt fp32 on the DISGAT shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from edgedisentangle_ssl_amd import ops_gemm  # noqa: E402
from kbench import timeit  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda")
for name, k, n, hb in (("P/Q   x^T gP   [256 x M] [M x 2048]", 256, 2048, 0), ("fuser f^T g   [2048 x M] [M x 256]", 2048, 256, 0),
                       ("MLP           [256 x M] [M x 256]", 256, 256, 0), ("heads 8 x     [256 x M] [M x 256]", 256, 256, 8)):
    if hb:
        a = torch.randn(M, hb, k, device=dev).permute(1, 0, 2)
        g = torch.randn(M, hb * n, device=dev).view(M, hb, n).permute(1, 0, 2)
        blas = lambda: torch.bmm(a.transpose(1, 2), g)
        flops = 2.0 * M * hb * k * n
    else:
        a = torch.randn(M, k, device=dev)
        g = torch.randn(M, n, device=dev)
        blas = lambda: a.t() @ g
        flops = 2.0 * M * k * n
    am, gm = ops_gemm.amax(a), ops_gemm.amax(g if not hb else g.permute(1, 0, 2).reshape(M, hb * n))
    t_b = timeit(blas, 5)
    t_k = timeit(lambda: ops_gemm._weight_grad(a, g, am, gm), 5)
    print(f"{name:40s} hipBLASLt {t_b:7.3f} ms {flops / t_b / 1e9:6.1f} TF   f16x3 split-K {t_k:7.3f} ms {flops / t_k / 1e9:6.1f} TF")
