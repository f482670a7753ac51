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
gen = torch.Generator(device=dev)
gen.manual_seed(1)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx, lab = sampling.sample_pairs(n, pos, gen)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"sample_pairs N={n} nnz={g.nnz}: M={idx.shape[1]} positives={int(lab.sum())} {dt * 1e3:.1f} ms", flush=True)
