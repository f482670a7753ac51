#!/usr/bin/env python3
"""T_iter with the three SSL losses on three HIP streams (they share weights and inputs and do not depend on one another in a
forward-only iteration) against the sequential form, same process, interleaved."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
o = bench.parse()
dev = torch.device("cuda")
args, enc, trainers, graph, x, lists = bench.build_workload(o, 0, 1, dev)
sup, dis, dif = trainers
(si, sl), (hi, hl), (ti, tl) = lists
data = (x, graph)
fns = (lambda: sup.loss(data, sl, [si]), lambda: dis.loss(data, [hl, tl], [hi, ti]), lambda: dif.loss(data))
streams = [torch.cuda.Stream() for _ in fns]


def sequential():
    with torch.no_grad():
        return sum(f() for f in fns)


def concurrent():
    cur = torch.cuda.current_stream()
    outs = []
    with torch.no_grad():
        for st, f in zip(streams, fns):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(f())
        for st in streams:
            cur.wait_stream(st)
    return sum(outs)


def timed(fn, n=5):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, float(r)


for _ in range(2):
    sequential()
    concurrent()
for rep in range(3):
    a, va = timed(sequential)
    b, vb = timed(concurrent)
    print(f"sequential {a:8.2f} ms (sum of losses {va:.6f})   three streams {b:8.2f} ms ({vb:.6f})   peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
