"""One rank's share of the multi-GPU configurations of BASELINE.json, on one GPU, through the real layer code
(layers.disga_heads, the trainers' losses) with the collectives replaced by local stand-ins:

  * configs[4] (8M nodes / 160M edges, 256-dim, 8 heads, gnn_type GCN, 8 GPUs): rank 0's 1M rows / ~20M entries of the
    8M-node graph - 8M-row gather tables (x 8 GB, Q 65 GB), column ids up to 8M - checked through size-independent
    properties: attention rows sum to one, the aux scorer on the rank's own edges reproduces the edge pass, chunk
    invariance, the GCN bias path (layers.py:404-407, 38-54), finite losses, peak memory under 200 GiB;
  * configs[3] (1M / 20M, 4 GPUs): rank 0 of DistGraph.shard(full, 0, 4) against the same rows of the unsharded run.

What the stand-ins replace is communication only (tests/test_parallel_gloo.py, test_gpu_parallel.py and test_gpu_rccl.py
cover that side); every kernel, the partition, the global-column indexing and the table sizes are the real ones."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _standin_collectives(monkeypatch, tables):
    """parallel.exchange -> the next table of `tables` with this rank's rows overwritten by the real local input;
    reductions -> identity (one rank's partial sums)."""
    from edgedisentangle_ssl_amd import parallel
    calls = {"n": 0}

    def exchange(x, g, edge_only, pipelined=False):
        if not (isinstance(g, parallel.DistGraph) and g.world > 1):
            return x, g
        t = tables[calls["n"] % len(tables)]
        calls["n"] += 1
        assert t.shape[0] == g.n_global and t.shape[1] == x.shape[1]
        t[g.row_start: g.row_start + g.n] = x
        return t, g

    monkeypatch.setattr(parallel, "exchange", exchange)
    monkeypatch.setattr(parallel, "all_reduce_sum", lambda t, g: t)
    monkeypatch.setattr(parallel, "all_reduce_max", lambda t, g: t)
    return calls


def test_configs4_one_rank_share_gcn(monkeypatch):
    import bench
    from edgedisentangle_ssl_amd import layers, ops, parallel
    dev = torch.device("cuda")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gnn_type", "GCN"])
    o = bench.parse()
    world = 8
    torch.cuda.reset_peak_memory_stats()
    a, enc, trainers, graph, x, lists = bench.build_workload(o, 0, world, dev)
    assert isinstance(graph, parallel.DistGraph) and graph.n_global == 8_000_000 and graph.n == 1_000_000
    assert int(graph.col.max()) >= 7_000_000                      # column ids really span the 8M-node graph
    gen = torch.Generator(device="cuda").manual_seed(17)
    table = torch.randn(graph.n_global, o.feat, device=dev, generator=gen)       # the other ranks' rows: stand-in values
    calls = _standin_collectives(monkeypatch, [table])
    heads_l1 = enc.attentions1
    with torch.no_grad():
        # (1) attention rows sum to one: identity GCN weights, zero bias, x = 1 everywhere -> every head output is elu(1) = 1
        saved = [(l.ag_layer.weight.detach().clone(), l.ag_layer.bias.detach().clone()) for l in heads_l1]
        table_one = torch.ones_like(table)
        monkeypatch.setattr(parallel, "exchange", lambda x_, g_, edge_only, pipelined=False: (table_one, g_))
        for l in heads_l1:
            l.ag_layer.weight.copy_(torch.eye(o.feat, device=dev))
            l.ag_layer.bias.zero_()
        h, e_list, _ = layers.disga_heads(heads_l1, torch.ones(graph.n, o.feat, device=dev), graph)
        fused = h.fused if h.fused is not None else h.planes.to_f32()
        assert float((fused - 1.0).abs().max()) < 5e-6
        # (2) GCN bias path: zero weights -> every head output is elu(bias) whatever the attention is
        for l in heads_l1:
            l.ag_layer.weight.zero_()
            l.ag_layer.bias.copy_(torch.linspace(-1.0, 1.0, o.feat, device=dev))
        h, _, _ = layers.disga_heads(heads_l1, x, graph)
        fused = h.fused if h.fused is not None else h.planes.to_f32()
        want = torch.nn.functional.elu(torch.linspace(-1.0, 1.0, o.feat, device=dev)).repeat(o.heads)
        assert float((fused - want).abs().max()) < 5e-6
        for l, (w_, b_) in zip(heads_l1, saved):
            l.ag_layer.weight.copy_(w_)
            l.ag_layer.bias.copy_(b_)
        del fused, h, table_one
        layers.clear_weight_cache(enc)
        _standin_collectives(monkeypatch, [table])
        # (3) the aux scorer on the rank's own edges (local row, GLOBAL column up to 8M) == the edge pass's raw scores
        sel = torch.sort(torch.randperm(graph.nnz, device=dev, generator=gen)[:3_000_000]).values
        pairs = torch.stack([graph.row[sel], graph.col[sel].long()]).contiguous()
        h1, e1, aux = layers.disga_heads(heads_l1, x, graph, [pairs])
        for hd in range(o.heads):
            ref = e1[hd][sel]
            assert float((aux[hd][0] - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
        # (4) chunk invariance of the split hub rows
        f1 = h1.fused if h1.fused is not None else h1.planes.to_f32()
        monkeypatch.setattr(ops, "CHUNK", {1: 1 << 30, 2: 1 << 30, 3: 1 << 30, 4: 1 << 30})
        graph._items.clear()
        h1n, e1n, _ = layers.disga_heads(heads_l1, x, graph)
        f1n = h1n.fused if h1n.fused is not None else h1n.planes.to_f32()
        assert all(torch.equal(p_, q_) for p_, q_ in zip(e1, e1n))
        assert float((f1 - f1n).abs().max()) <= 2e-5 * max(1.0, float(f1.abs().max()))
        monkeypatch.undo()
        graph._items.clear()
        del f1, f1n, h1, h1n, e1, e1n, aux
        # (5) the whole T_iter step of this rank (three SSL losses on its pair lists): finite, within the memory envelope
        _standin_collectives(monkeypatch, [table, torch.randn(graph.n_global, o.feat, device=dev, generator=gen)])
        loss = bench.one_step(o, enc, trainers, graph, x, lists)
        assert torch.isfinite(loss).all()
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    assert peak < 200.0, f"peak {peak:.1f} GiB"
    print(f"configs[4] rank share: peak {peak:.1f} GiB, {calls['n']} exchanges served")


def test_configs3_rank0_of_4_equals_unsharded_rows(monkeypatch):
    from types import SimpleNamespace
    from edgedisentangle_ssl_amd import DISGAT, FuseLayer, parallel, synth
    dev = torch.device("cuda")
    n, e, f, H = 1_000_000, 20_000_000, 256, 8
    a = SimpleNamespace(gnn_type="AT", att=3, nhead=H, nhid=f, size=f, residue=False, residue_type=0, fuse_no_relu=False,
                        dropout=0.0, cls_layer=2, constrain_layer=0, sparse=True, model="DISGAT", dis_type=1, lr=0.01,
                        weight_decay=5e-4)
    torch.manual_seed(0)
    enc = DISGAT(a, nfeat=f, nhid=f, nclass=f, nheads=H, dropout=0.0).to(dev).eval()
    fus = [FuseLayer(a, H, nfeat=f).to(dev).eval(), FuseLayer(a, H, nfeat=f).to(dev).eval()]
    full = synth.powerlaw_graph(n, e, dev)
    x = synth.features(n, f, dev)
    with torch.no_grad():
        ref = enc.get_em(x, full, fus)
        ref = [t.clone() for t in ref]
    shard = parallel.DistGraph.shard(full, 0, 4)
    lo, hi = shard.row_start, shard.row_start + shard.n
    assert lo == 0 and abs(shard.nnz - full.nnz / 4) < 0.02 * full.nnz and shard.n_global == n
    # the other ranks' rows of each layer's input: what they would have computed = the unsharded run's values
    _standin_collectives(monkeypatch, [x.clone(), ref[0].clone()])
    with torch.no_grad():
        got = enc.get_em(x[lo:hi].contiguous(), shard, fus)
    for l in range(2):
        r_ = ref[l][lo:hi]
        assert got[l].shape == r_.shape
        assert float((got[l] - r_).abs().max()) <= 1e-6 * max(1.0, float(r_.abs().max())), l
