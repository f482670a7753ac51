#!/usr/bin/env python3
"""split-bf16 GEMM vs hipBLASLt fp32 on the DISGAT shapes (M = 1e6 rows)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import ops_gemm  # noqa: E402
from kbench import timeit  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda")
cases = [("P/Q  x[M,256] @ [256,2048]", (M, 256), (256, 2048), None),
         ("fuse [M,2048] @ [2048,256]", (M, 2048), (2048, 256), None),
         ("MLP  [M,256] @ [256,256]", (M, 256), (256, 256), None),
         ("heads Z[M,8,256] @ [8,256,256] -> [M,2048]", None, None, 8)]
for name, ashape, wshape, hb in cases:
    if hb:
        z = torch.randn(M, hb, 256, device=dev)
        a = z.permute(1, 0, 2)
        w = torch.randn(hb, 256, 256, device=dev) * 0.05
        flops = 2.0 * M * hb * 256 * 256
        blas = lambda: torch.bmm(a, w)
    else:
        a = torch.randn(*ashape, device=dev)
        w = torch.randn(*wshape, device=dev) * 0.05
        flops = 2.0 * ashape[0] * ashape[1] * wshape[1]
        blas = lambda: a @ w
    res = {}
    ref = None
    for md in ("blas", "split6", "f16x3", "f16x3+amax"):
        os.environ["DISGAT_GEMM"] = md.split("+")[0]
        am = ops_gemm.amax(a) if md == "f16x3" else None          # "f16x3": amax precomputed; "+amax": pass included
        fn = blas if md == "blas" else (lambda: ops_gemm._forward(a, w, None, None, 0, 0.0, am))
        ms = timeit(fn, 5)
        res[md] = (ms, flops / ms / 1e9)
        if M <= 200_000:                                           # error vs fp64 on small runs
            out = fn()
            out = out if not hb else out.view(M, hb, -1).permute(1, 0, 2) if md != "blas" else out
            if ref is None:
                ref = (a.double() @ w.double()) if not hb else torch.bmm(a.double(), w.double())
            res[md] += (float((out.double() - ref).abs().max() / ref.abs().max()),)
    print(f"{name:44s} " + "  ".join(f"{k}: {v[0]:6.3f} ms {v[1]:5.1f} TF" + (f" err {v[2]:.1e}" if len(v) > 2 else "")
                                       for k, v in res.items()))
