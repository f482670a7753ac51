#!/usr/bin/env python3
"""Forward+backward+Adam timing of the three SSL train_steps on a synthetic power-law graph."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=200_000)
ap.add_argument("--edges", type=int, default=4_000_000)
ap.add_argument("--feat", type=int, default=256)
ap.add_argument("--att", type=int, default=3)
ap.add_argument("--gnn_type", default="AT")
ap.add_argument("--dropout", type=float, default=0.1)
ap.add_argument("--skip-unused", action="store_true", help="DISGAT.skip_unused as main.run sets it (no discarded layer-2 aggregation)")
ap.add_argument("--sampled", choices=("none", "exact", "padded"), default="none",
                help="also time the trainers' own train_step()s: lists drawn every step, exact length or fixed capacity (padded)")
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
sys.argv = ["bench.py", "--nodes", str(a.nodes), "--edges", str(a.edges), "--feat", str(a.feat), "--att", str(a.att),
            "--gnn_type", a.gnn_type]
o = bench.parse()
dev = torch.device("cuda")
args, enc, (sup, dis, dif), graph, x, lists = bench.build_workload(o, 0, 1, dev)
enc.skip_unused = a.skip_unused
for m in list(enc.modules()):
    if hasattr(m, "dropout"):
        m.dropout = a.dropout
(si, sl), (hi, hl), (ti, tl) = lists
data = (x, graph)


def fwd_only():
    with torch.no_grad():
        return sup.loss(data, sl, [si]) + dis.loss(data, [hl, tl], [hi, ti]) + dif.loss(data)


PEAKS = {}


def train():
    out = []
    for nm, tr, fn in (("sup", sup, lambda: sup.loss(data, sl, [si])), ("dis", dis, lambda: dis.loss(data, [hl, tl], [hi, ti])),
                       ("dif", dif, lambda: dif.loss(data))):
        tr._begin_step()
        if os.environ.get("TRAIN_BENCH_PEAKS") == "1":
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            loss = fn()
            torch.cuda.synchronize()
            PEAKS[nm + " end of forward (held)"] = torch.cuda.memory_allocated() / 2 ** 30
            PEAKS[nm + " forward peak"] = torch.cuda.max_memory_allocated() / 2 ** 30
            torch.cuda.reset_peak_memory_stats()
            tr._finish_step(loss, graph)
            torch.cuda.synchronize()
            PEAKS[nm + " backward peak"] = torch.cuda.max_memory_allocated() / 2 ** 30
        else:
            loss = fn()
            tr._finish_step(loss, graph)
        out.append(loss.detach())
    return out


cases = [("forward only (eval)", fwd_only), ("train (fwd+bwd+Adam, dropout %.2f)" % a.dropout, train)]
if a.sampled != "none":
    from edgedisentangle_ssl_amd import sampling, synth
    dis.get_label_all(x, graph, synth.node_labels(graph.n, dev))
    sampling.PADDED_LISTS = a.sampled == "padded"

    def train_sampled():
        return [sup.train_step(data, graph), dis.train_step(data), dif.train_step(data)]
    cases.append((f"train_step()s with sampled lists ({a.sampled})", train_sampled))
for name, fn in cases:
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = a.iters
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"{name:44s} {dt * 1e3:9.2f} ms/iter  {graph.nnz / dt:.3e} edges/s   peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
if os.environ.get("TRAIN_BENCH_PEAKS") == "1":
    for k, v in PEAKS.items():
        print(f"  {k:32s} {v:7.1f} GiB")
