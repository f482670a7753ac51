#!/usr/bin/env python3
"""Runs ONE dense contraction of the DISGAT path a few times (for rocprofv3 PMC passes), M = 1,000,000 rows:
  tools/gemm_one.py pq       x [M,256] @ [256,2048]          fp32 operand, A-stationary kernel (P / Q score operands)
  tools/gemm_one.py proj     Z planes [M,8,256] @ [8,256,256] -> ELU -> head planes      (disgat_gemm_planes)
  tools/gemm_one.py fuser    head planes [M,2048] @ [2048,256] + bias, leaky ReLU -> fp32 (disgat_gemm_planes)
  tools/gemm_one.py b2b      Z planes [M,8,256] @ [8,256,256] -> ELU -> @ [2048,256] + bias, leaky ReLU -> fp32, one launch (disgat_proj_fuse)
  tools/gemm_one.py logits   head planes [M,8,256] @ [256,256] + shared, leaky ReLU -> @ [256,8] + bias: [8M,8] logits, one launch (disgat_gemm_planes_logits)
  tools/gemm_one.py K N      fp32 operand [M,K] @ [K,N] on the fp32-input kernels (round-2 form)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import ops_gemm as og  # noqa: E402

M = 1_000_000
what = sys.argv[1]
if what == "proj":
    a = torch.randn(M, 8, 256, device="cuda").permute(1, 0, 2)
    w = torch.randn(8, 256, 256, device="cuda") * 0.05
    ap, wr = og.split_planes(a), og.presplit_rm(w)
    bound = (ap.bound * w.abs().sum(1).max()).reshape(1)
    fn = lambda: og.linear_planes(ap, wr, 256, None, None, og.ACT_ELU, 0.0, False, bound)      # noqa: E731
elif what == "b2b":
    a = torch.randn(M, 8, 256, device="cuda").permute(1, 0, 2)
    w1 = torch.randn(8, 256, 256, device="cuda") * 0.05
    w2 = torch.randn(2048, 256, device="cuda") * 0.02
    b = torch.randn(256, device="cuda")
    ap, wch = og.split_planes(a), og.presplit_b2b(w1, w2)
    del a
    bound = torch.clamp(ap.bound * w1.abs().sum(1).max() * 1.001, min=1.0).reshape(1)
    fn = lambda: og.proj_fuse(ap, wch, None, b, bound, 256, 256, og.ACT_LEAKY, 0.01)           # noqa: E731
elif what == "logits":
    a = torch.randn(M, 8, 256, device="cuda").permute(1, 0, 2)
    w = (torch.randn(256, 256, device="cuda") * 0.05).unsqueeze(0).expand(8, 256, 256)     # DifHead's classifier: the same weight for every head
    shared = torch.randn(M, 256, device="cuda")
    lin2 = torch.nn.Linear(256, 8).cuda()
    ap, wr, w2 = og.split_planes(a), og.presplit_rm(w), og.presplit_logits(lin2.weight, lin2.bias)
    del a
    mid = (ap.bound * w[0].abs().sum(0).max() * 1.001 + shared.abs().max()).reshape(1)
    fn = lambda: og.linear_planes_logits(ap, wr, None, shared, og.ACT_LEAKY, 0.01, mid, w2)                 # noqa: E731
elif what == "fuser":
    h = torch.randn(M, 2048, device="cuda")
    w = torch.randn(2048, 256, device="cuda") * 0.02
    b = torch.randn(256, device="cuda")
    hp, wr = og.split_planes(h), og.presplit_rm(w)
    fn = lambda: og.linear_planes(hp, wr, 256, b, None, og.ACT_LEAKY, 0.01)                     # noqa: E731
else:
    K, N = (256, 2048) if what == "pq" else (int(sys.argv[1]), int(sys.argv[2]))
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(K, N, device="cuda") * 0.05
    am, ws = og.amax(a), og.presplit(w)
    fn = lambda: og._forward(a, w, None, None, 0, 0.0, am, ws)                                  # noqa: E731
for _ in range(4):
    fn()
torch.cuda.synchronize()
