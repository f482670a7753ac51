#!/usr/bin/env python3
"""Probe: does processing the aux pairs in column-bucket-major order (bucket = c // C, row-sorted inside) make the
Q[c] gathers hit the memory-side cache?  Times disgat_aux_score on the row-major list and on bucketed permutations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from edgedisentangle_ssl_amd import ops, synth  # noqa: E402
from kbench import timeit  # noqa: E402

dev = torch.device("cuda")
n, f, H = 1_000_000, 256, 8
m = 66_546_613
pairs, _ = synth.uniform_pairs(n, m, dev)
rowop = torch.randn(n, H * f, device=dev)
colop = torch.randn(n, H * f, device=dev)
a = torch.randn(H * f, device=dev)
b = ops.aux_algorithmic_bytes(3, n, m, H, f, f)
ms = timeit(lambda: ops.aux_forward(3, H, f, f, pairs, n, None, rowop, colop, a, 0, H), 3)
print(f"row-major: {ms:8.3f} ms  {b / ms / 1e6:8.1f} GB/s", flush=True)
ref = ops.aux_forward(3, H, f, f, pairs, n, None, rowop, colop, a, 0, H)
for nb in (16, 32, 64, 128, 256, 512):
    C = (n + nb - 1) // nb
    torch.cuda.synchronize()
    key = (pairs[1] // C) * (n * n) + pairs[0] * n + pairs[1]
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    perm = torch.sort(key).indices
    p2 = pairs[:, perm].contiguous()
    e.record(); torch.cuda.synchronize()
    t_sort = s.elapsed_time(e)
    del key
    ms = timeit(lambda: ops.aux_forward(3, H, f, f, p2, n, None, rowop, colop, a, 0, H), 3)
    out = ops.aux_forward(3, H, f, f, p2, n, None, rowop, colop, a, 0, H)
    ok = torch.equal(out, ref[:, perm])
    print(f"{nb:4d} column buckets (C={C}): {ms:8.3f} ms  {b / ms / 1e6:8.1f} GB/s algorithmic-equivalent; reorder {t_sort:.1f} ms; same scores {ok}", flush=True)
    del perm, p2, out
