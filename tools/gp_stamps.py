#!/usr/bin/env python3
"""Where a k-step of disgat_gemm_planes spends its cycles (s_memtime stamps, DISGAT_PL_DEBUG=32)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import _lib, ops_gemm as og  # noqa: E402
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
h = torch.randn(M, 2048, device=dev)
wf = torch.randn(2048, 256, device=dev) * 0.02
amh = og.amax(h)
wfr = og.presplit_rm(wf)
hp = og.split_planes(h, amh)
names = ["wait", "barrier", "frag+issue", "mfma", "epilogue", "unit-setup"]
for ring in sys.argv[1:] or ["52", "43"]:
    os.environ["DISGAT_PL_RING"] = ring
    for what, fn, steps in (("proj", lambda: og.linear_planes(ap, wr, 256, None, None, 1, 0.0, False, bound), 62504 * 8),
                            ("fuser", lambda: og.linear_planes(hp, wfr, 256, None, None, 2, 0.01, True, None), 7813 * 64)):
        os.environ["DISGAT_PL_DEBUG"] = "0"
        t0 = timeit(fn, 5)
        os.environ["DISGAT_PL_DEBUG"] = "32"
        fn()
        _lib.call("disgat_debug_stamps", None, 1)
        fn()
        buf = (ctypes.c_ulonglong * 16)()
        _lib.call("disgat_debug_stamps", buf, 1)
        v = list(buf)
        # s_memtime ticks at 100 MHz on gfx9?  report raw ticks per k-step (sum over 256 blocks / steps)
        for role in (0, 1):
            tot = sum(v[role * 8:role * 8 + 6])
            per = "  ".join(f"{names[k]} {v[role * 8 + k] / steps:8.1f}" for k in range(6))
            print(f"ring={ring} {what} ({t0:.3f} ms) {'A' if role == 0 else 'B'}-wave ticks per k-step: {per}   total {tot / steps:8.1f}", flush=True)
