#!/usr/bin/env python3
"""In-kernel phase stamps of gemm_f16x3_rs_kernel on the P/Q shape (needs DISGAT_HIPCC_FLAGS=-DRS_DIAG=1 and DISGAT_RS_DEBUG=32[+ablation bits])."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import _lib, ops_gemm as og  # noqa: E402

M, K, N = 1_000_000, 256, 2048
a = torch.randn(M, K, device="cuda")
w = torch.randn(K, N, device="cuda") * 0.05
am, ws = og.amax(a), og.presplit(w)
for _ in range(2):
    og._forward(a, w, None, None, 0, 0.0, am, ws)
_lib.call("disgat_debug_stamps_rs", None, 1)
reps = 4
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    og._forward(a, w, None, None, 0, 0.0, am, ws)
e.record()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
_lib.call("disgat_debug_stamps_rs", buf, 1)
blocks = (M + 255) // 256
chunks = N // 32
names = ["vmcnt wait", "barrier", "DMA issue", "seed", "k-steps", "stores+tail", "stores prev (late)", "retire VALU"]
print(f"{s.elapsed_time(e) / reps:.3f} ms per launch; s_memtime ticks per chunk and wave (100 MHz clock: x ~21-24 for shader cycles)")
for role in range(2):
    tot = sum(buf[role * 8 + i] for i in range(8))
    print(("early wave 0" if role == 0 else "late wave 4 ") + "  " + "  ".join(
        f"{names[i]} {buf[role * 8 + i] / (reps * blocks * chunks):7.2f}" for i in range(8)) + f"   total {tot / (reps * blocks * chunks):7.2f}")
