#!/usr/bin/env python3
"""In-kernel phase stamps of proj_fuse_kernel at the C4 shape (needs a DISGAT_HIPCC_FLAGS=-DBB_DIAG=32[+ablation bits] build: tools/b2b_ablate.sh)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import _lib, ops_gemm as og  # noqa: E402

m, H, F = 1_000_000, 8, 256
g = torch.Generator(device="cuda").manual_seed(0)
z = torch.randn(m, H, F, device="cuda", generator=g)
w1 = torch.randn(H, F, F, device="cuda", generator=g) * 0.09
w2 = torch.randn(H * F, F, device="cuda", generator=g) * 0.02
b2 = torch.randn(F, device="cuda", generator=g) * 0.1
zp = og.split_planes(z.permute(1, 0, 2))
del z
bound = torch.clamp(zp.bound * w1.abs().sum(1).max() * 1.001, min=1.0).reshape(1)
wch = og.presplit_b2b(w1, w2)
run = lambda: og.proj_fuse(zp, wch, None, b2, bound, F, F, og.ACT_LEAKY, 0.01)
for _ in range(2):
    run()
_lib.call("disgat_debug_stamps_b2b", None, 1)
reps = 4
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    run()
e.record()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
_lib.call("disgat_debug_stamps_b2b", buf, 1)
steps = reps * ((m + 127) // 128) * H * (F // 32)
names = ["vmcnt wait", "barrier", "seed+reads", "GEMM 1", "ELU/split", "GEMM 2+tail"]
tot = sum(buf[i] for i in range(6))
print(f"{s.elapsed_time(e) / reps:.3f} ms per launch; s_memtime ticks per step of wave 0: " + "  ".join(
    f"{names[i]} {buf[i] / steps:7.1f}" for i in range(6)) + f"   total {tot / steps:7.1f}")
