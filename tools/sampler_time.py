#!/usr/bin/env python3
"""Wall time of the O(E) pair sampler at bench scale: tools/sampler_time.py [nodes edges]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import sampling, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
e = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
dev = torch.device("cuda")
g = synth.powerlaw_graph(n, e, dev)
pos = sampling.flat_edges(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
smp = sampling.PairSampler(n, pos, seed=1)
torch.cuda.synchronize()
print(f"PairSampler set-up (item table, once per positive set): {(time.perf_counter() - t0) * 1e3:.1f} ms, {smp.n_items} items, "
      f"{smp.ipw} per wave, capacity {smp.capacity}", flush=True)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx, lab = smp.sample()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"sample N={n} nnz={g.nnz}: M={idx.shape[1]} positives={int(lab.sum())} {dt * 1e3:.2f} ms wall", flush=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for form in ("sample", "sample_static"):
    fn = getattr(smp, form)
    fn()
    times = []
    for rep in range(10):
        ev[0].record()
        smp._plan(smp.capacity if form == "sample_static" else (1 << 62), None)
        ev[1].record()
        out = fn()
        ev[2].record()
        torch.cuda.synchronize()
        times.append((ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])))
    plan = sorted(t[0] for t in times)[len(times) // 2]
    whole = sorted(t[1] for t in times)[len(times) // 2]
    print(f"{form}: plan (count + scan) {plan:.3f} ms, plan + emit {whole:.3f} ms (device time, median of 10)", flush=True)
print("events (overflow, clamped):", smp.events())
