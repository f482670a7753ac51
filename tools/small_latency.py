#!/usr/bin/env python3
"""Latency of the DISGAT path on the small real graphs (BASELINE configs[0]/[1]): get_em and
T_iter per call, eager vs replayed as one HIP graph (torch.cuda.CUDAGraph capture)."""
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs_common as ic  # noqa: E402
from edgedisentangle_ssl_amd import DISGAT, data_load, pretrainer  # noqa: E402
from edgedisentangle_ssl_amd.graph import graph_of  # noqa: E402


def bench(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for name in ("cora", "chameleon", "cora_full"):
    adj, feats, labels = data_load.load_fixture(os.path.join(ROOT, "tests/golden", f"data_{name}.npz"))
    n = adj.shape[0]
    if feats is None:
        feats = ic.features(51, n, 64, "cora_surrogate")
    dev = torch.device("cuda")
    a = SimpleNamespace(gnn_type="AT", att=3, nhead=8, nhid=64, size=feats.shape[1], residue=False, residue_type=0,
                        fuse_no_relu=False, dropout=0.0, cls_layer=2, constrain_layer=0, sparse=True, model="DISGAT",
                        dis_type=1, lr=0.01, weight_decay=5e-4)
    enc = DISGAT(a, nfeat=a.size, nhid=64, nclass=64, nheads=8, dropout=0.0).to(dev).eval()
    sup = pretrainer.SupEdgeTrainer(a, enc, 1.0)
    dis = pretrainer.GeneratedEdgeTrainer(a, enc, 1.0)
    dif = pretrainer.DifHeadTrainer(a, enc, 1.0)
    x, adj, labels = feats.to(dev), adj.to(dev), labels.to(dev)
    g = graph_of(adj)
    dis.get_label_all(x, g, labels)
    sl, si = sup.sample_train(g)
    dl, di = dis.sample_train()
    fus = [sup.fuse1, sup.fuse2]

    def fwd():
        with torch.no_grad():
            return enc.get_em(x, g, fus)

    def titer():
        with torch.no_grad():
            return sup.loss((x, g), sl, si) + dis.loss((x, g), dl, di) + dif.loss((x, g))

    res = {"get_em": bench(fwd), "T_iter": bench(titer)}
    # one HIP graph per call: static shapes (graph, pair lists resident)
    for nm, fn in (("get_em", fwd), ("T_iter", titer)):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        cg = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(cg):
                out = fn()
            res[nm + "_graph"] = bench(cg.replay)
        except Exception as e:  # noqa: BLE001
            res[nm + "_graph"] = f"capture failed: {type(e).__name__}: {str(e)[:120]}"
    print(name, f"N={n} E={g.nnz} M_sup={si[0].shape[1]}", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}, "ms")
