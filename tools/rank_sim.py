#!/usr/bin/env python3
"""What ONE rank of an N-GPU weak-scaling run executes (bench.py's sharded workload), with the
collectives stubbed (all-gather = local repeat, all-reduce = identity): checks that the per-rank
working set fits and measures the compute side of the scaling loss (the redundant column-operand
GEMM and the larger gather table).  Communication time is NOT included."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from edgedisentangle_ssl_amd import parallel  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sys.argv = ["bench.py"] + sys.argv[2:]
o = bench.parse()
dev = torch.device("cuda")


def _fake_gather(x, g):
    """N_global rows with this rank's own in place (other ranks' rows: copies - values do not matter for timing)."""
    if not (isinstance(g, parallel.DistGraph) and g.world > 1):
        return x
    reps = -(-g.n_global // x.shape[0])
    return x.repeat(reps, 1)[: g.n_global].contiguous()


parallel.all_gather_rows = _fake_gather
parallel.exchange = lambda x, g, edge_only, pipelined=False: (_fake_gather(x, g), g)
parallel.all_reduce_max = lambda t, g: t
parallel.all_reduce_sum = lambda t, g: t
a, enc, trainers, graph, x, lists = bench.build_workload(o, 0, world, dev)
for _ in range(1):
    bench.one_step(o, enc, trainers, graph, x, lists)
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
t = time.perf_counter()
n = 2
for _ in range(n):
    last = bench.one_step(o, enc, trainers, graph, x, lists)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
print(f"world={world}: one rank's step {dt * 1e3:.1f} ms (no comm), local nnz {graph.nnz}, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, loss finite {bool(torch.isfinite(last))}")
