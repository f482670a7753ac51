#!/usr/bin/env python3
"""Kernel micro-benchmark: times the HIP launchers directly on C4-shaped random operands
(no model, no GEMMs), reporting algorithmic GB/s per launch.  For kernel iteration only."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import ops, synth  # noqa: E402


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=20_000_000)
    ap.add_argument("--feat", type=int, default=256)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--att", type=int, nargs="+", default=[3])
    ap.add_argument("--what", nargs="+", default=["edge", "aux", "auxhalf"])
    o = ap.parse_args()
    dev = torch.device("cuda")
    n, f, H = o.nodes, o.feat, o.heads
    g = synth.powerlaw_graph(n, o.edges, dev)
    x = torch.randn(n, f, device=dev)
    m = (10 * g.nnz) // 3
    pairs, _ = synth.uniform_pairs(n, m, dev)
    pairs_q, _ = synth.uniform_pairs(n, m // 4, dev, seed=100)
    for att in o.att:
        if att == 3:
            rowop = torch.randn(n, H * f, device=dev)
            colop = torch.randn(n, H * f, device=dev)
            a = torch.randn(H * f, device=dev)
        elif att == 2:
            rowop, colop, a = torch.randn(n, H * f, device=dev), None, None
        else:
            rowop, colop, a = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev), None
        if "edge" in o.what:
            ms = timeit(lambda: ops.edge_forward(g, att, H, f, f, x, rowop, colop, a, False))
            b = ops.edge_algorithmic_bytes(att, n, g.nnz, H, f, f)
            print(f"edge_fwd att{att}: {ms:8.3f} ms  {b / ms / 1e6:8.1f} GB/s  ({b / 1e9:.1f} GB)")
        if "aux" in o.what:
            ms = timeit(lambda: ops.aux_forward(att, H, f, f, pairs, n, x, rowop, colop, a, 0, H), 3)
            b = ops.aux_algorithmic_bytes(att, n, m, H, f, f)
            print(f"aux att{att} all heads M={m}: {ms:8.3f} ms  {b / ms / 1e6:8.1f} GB/s")
        if "auxhalf" in o.what:
            for lo, hi, p in ((0, H // 2, pairs_q), (H // 2, H, pairs)):
                mm = p.shape[1]
                ms = timeit(lambda: ops.aux_forward(att, H, f, f, p, n, x, rowop, colop, a, lo, hi), 3)
                b = ops.aux_algorithmic_bytes(att, n, mm, hi - lo, f, f)
                print(f"aux att{att} heads [{lo},{hi}) M={mm}: {ms:8.3f} ms  {b / ms / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
