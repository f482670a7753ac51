#!/usr/bin/env python3
"""Same-process A/B of disgat_gemm_planes switches (DISGAT_PL_DEBUG bits) on the projection / fuser shapes, interleaved rounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import ops_gemm as og  # noqa: E402
from kbench import timeit  # noqa: E402

dev = torch.device("cuda")
M = 1_000_000
a = torch.randn(M, 8, 256, device=dev).permute(1, 0, 2)
w = torch.randn(8, 256, 256, device=dev) * 0.05
wr = og.presplit_rm(w)
ap = og.split_planes(a)
bound = (ap.bound * w.abs().sum(1).max()).reshape(1)
h = torch.randn(M, 2048, device=dev)
wf = torch.randn(2048, 256, device=dev) * 0.02
wfr = og.presplit_rm(wf)
hp = og.split_planes(h)
vals = sys.argv[1:] or ["0", "64"]
res = {v: ([], []) for v in vals}
for rnd in range(4):
    for v in vals:
        os.environ["DISGAT_PL_DEBUG"] = v
        res[v][0].append(timeit(lambda: og.linear_planes(ap, wr, 256, None, None, 1, 0.0, False, bound), 5))
        res[v][1].append(timeit(lambda: og.linear_planes(hp, wfr, 256, None, None, 2, 0.01, True, None), 5))
for v in vals:
    p, f = sorted(res[v][0]), sorted(res[v][1])
    print(f"dbg={v}: proj median {p[len(p) // 2]:.3f} (min {p[0]:.3f})  fuser median {f[len(f) // 2]:.3f} (min {f[0]:.3f})", flush=True)
