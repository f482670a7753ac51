#!/usr/bin/env python3
"""Ablations of disgat_gemm_planes on the projection / fuser shapes (DISGAT_PL_DEBUG bits) and an A-stride probe."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import ops_gemm as og  # noqa: E402
from kbench import timeit  # noqa: E402

dev = torch.device("cuda")
M = 1_000_000
z = torch.randn(M, 8, 256, device=dev)
a = z.permute(1, 0, 2)
w = torch.randn(8, 256, 256, device=dev) * 0.05
am = og.amax(a)
wr = og.presplit_rm(w)
ap = og.split_planes(a, am)
bound = (am * w.abs().sum(1).max()).reshape(1)
# head-contiguous planes [8][M][256]: row stride 512 B instead of 4 KB
hc = og.Planes(ap.hi.contiguous(), ap.lo.contiguous(), am)
h = torch.randn(M, 2048, device=dev)
wf = torch.randn(2048, 256, device=dev) * 0.02
amh = og.amax(h)
wfr = og.presplit_rm(wf)
hp = og.split_planes(h, amh)
for ring in ("52", "52i", "43", "43i", "32", "32i"):
    os.environ["DISGAT_PL_RING"] = ring
    for dbg in (0, 1, 16):
        os.environ["DISGAT_PL_DEBUG"] = str(dbg)
        t1 = timeit(lambda: og.linear_planes(ap, wr, 256, None, None, 1, 0.0, False, bound), 5)
        t3 = timeit(lambda: og.linear_planes(hp, wfr, 256, None, None, 2, 0.01, True, None), 5)
        print(f"ring={ring} dbg={dbg} (1 no epilogue, 16 no epilogue math): proj planes->planes {t1:6.3f} ms | fuser {t3:6.3f} ms", flush=True)
