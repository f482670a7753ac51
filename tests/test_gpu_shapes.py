"""Shape sweep of the fused layer (all template instantiations the host can select) against the
float64 oracle on a small random graph: odd head counts (padded to a power of two), F_in not a
multiple of 4 / wider than 256 (two register tiles), every F_out padding class of att 3, H = 16,
forward outputs, raw scores, aux scores with head ranges, and gradients of everything.  No case is skipped."""
import numpy as np
import pytest
import torch

import inputs_common as ic
import kink
from test_gpu_parity import close, dev  # noqa: F401

pytestmark = pytest.mark.gpu

CASES = [
    # (H, F_in, F_out, att, gnn)
    (1, 16, 8, 3, "AT"), (2, 30, 20, 3, "SAGE"), (3, 33, 48, 3, "GCN"), (4, 64, 100, 3, "AT"),
    (6, 40, 70, 3, "SAGE"), (8, 96, 256, 3, "AT"), (16, 24, 128, 3, "GCN"), (16, 200, 12, 3, "AT"),
    (2, 300, 64, 3, "AT"), (8, 260, 32, 3, "SAGE"), (2, 512, 16, 3, "GCN"),
    (1, 10, 10, 1, "AT"), (5, 300, 33, 1, "SAGE"), (16, 130, 7, 1, "GCN"), (8, 256, 64, 1, "AT"),
    (1, 12, 9, 2, "GCN"), (3, 70, 50, 2, "AT"), (8, 300, 40, 2, "SAGE"), (16, 100, 16, 2, "AT"), (4, 512, 8, 2, "GCN"),
    # more heads / wider heads than one launch covers: head groups
    (24, 32, 40, 3, "AT"), (8, 64, 512, 3, "SAGE"), (4, 48, 1024, 3, "GCN"), (40, 20, 6, 1, "AT"), (20, 36, 12, 2, "SAGE"), (16, 257, 16, 2, "AT"), (32, 500, 8, 2, "GCN"),
    # wider than the register tile: aggregated in column slices (att 1 / 3 only)
    (4, 1433, 16, 3, "AT"), (8, 700, 32, 3, "SAGE"), (16, 300, 8, 1, "GCN"), (2, 1100, 24, 1, "AT"),
    # att 2 wider than its register tile (F_in > 512, e.g. Cora's raw 1 433-wide bag of words with --origin_feat):
    # scored through the per-head projections h = x W (kernel code 4), x aggregated in column slices
    (4, 1433, 16, 2, "AT"), (8, 700, 32, 2, "SAGE"), (3, 520, 70, 2, "GCN"), (20, 600, 24, 2, "AT"),
    # a single head wider than one launch scores (1024 features): feature slices, partial scores added through e_in
    (2, 40, 2048, 3, "AT"), (1, 24, 1500, 3, "GCN"), (3, 600, 1100, 2, "SAGE"), (5, 300, 1030, 3, "SAGE"),
]


def small_graph(n=96, seed=5):
    g = np.random.Generator(np.random.PCG64(seed))
    deg = np.minimum(n - 1, (g.pareto(1.2, n) * 3 + 1).astype(int))
    rows = np.repeat(np.arange(n), deg)
    cols = g.integers(0, n, rows.shape[0])
    hub = np.stack([np.full(n, 7), np.arange(n)])                     # one full row -> split path with chunk 16
    idx = np.concatenate([np.stack([rows, cols]), hub, np.stack([np.arange(n - 2), np.arange(n - 2)])], 1)
    return torch.from_numpy(idx.astype(np.int64)), n


@pytest.mark.parametrize("H,f_in,f_out,att,gnn", CASES)
def test_layer_shapes_forward_backward(dev, H, f_in, f_out, att, gnn, monkeypatch):
    import edgedisentangle_ssl_amd as pkg
    from edgedisentangle_ssl_amd import ops
    from oracle import disgat_oracle as orc
    monkeypatch.setattr(ops, "CHUNK", {1: 16, 2: 16, 3: 16, 4: 16})
    idx, n = small_graph()
    ci = ic.coalesced_index_set(idx, n)
    adj = torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1]), (n, n)).to(dev)
    aux = [ic.aux_pairs(8, n, 333, "s0"), ic.aux_pairs(9, n, 97, "s1")]
    lo1, hi1 = (0, max(1, H // 2))
    ranges = [None, (lo1, hi1)]
    layers = [ic.load_params(pkg.DisGALayer(f_in, f_out, dropout=0.0, alpha=0.1, att_type=att, gnn_type=gnn), 500 + h).to(dev).eval()
              for h in range(H)]
    x = ic.features(3, n, f_in) * 0.5
    xg = x.to(dev).requires_grad_(True)
    recorded = []
    with kink.record_operands(recorded):
        heads, e_list, aux_out = pkg.disga_heads(layers, xg, adj, [a.to(dev) for a in aux], ranges)
    # att 3: the oracle's gradient takes the kernels' side wherever a leaky-ReLU argument is within 1e-5 of the kink
    # (tests/kink.py); every forward comparison below is independent of it
    pins = kink.Pins(recorded, H, f_out, [(ci[0], ci[1])] + [(a_[0], a_[1]) for a_ in aux]) if att == 3 else None

    gen = np.random.Generator(np.random.PCG64(11))
    wh = torch.from_numpy(gen.standard_normal((H, n, f_out)))
    we = torch.from_numpy(gen.standard_normal((H, ci.shape[1])) * 0.1)
    wa0 = torch.from_numpy(gen.standard_normal((H, 333)) * 0.1)
    wa1 = torch.from_numpy(gen.standard_normal((H, 97)) * 0.1)
    loss = 0.0
    for h in range(H):
        loss = loss + (heads[h].double() * wh[h].to(dev)).sum() + (e_list[h][:, 0].double() * we[h].to(dev)).sum()
        loss = loss + (aux_out[h][0][:, 0].double() * wa0[h].to(dev)).sum()
        if lo1 <= h < hi1:
            loss = loss + (aux_out[h][1][:, 0].double() * wa1[h].to(dev)).sum()
        else:
            assert aux_out[h][1] is None
    loss.backward()

    xc = x.double().requires_grad_(True)
    ref_loss = 0.0
    sds = []
    for h, lay in enumerate(layers):
        sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in lay.state_dict().items()}
        sds.append(sd)
        with kink.pinned_oracle(pins):
            ho, e, au = orc.disga_layer(xc, ci, sd, att, gnn, aux)
        close(heads[h], ho.detach(), what=f"head {h}")
        close(e_list[h][:, 0], e.detach()[:, 0], what=f"edge_e {h}")
        close(aux_out[h][0][:, 0], au[0].detach()[:, 0], what=f"aux0 {h}")
        ref_loss = ref_loss + (ho * wh[h]).sum() + (e[:, 0] * we[h]).sum() + (au[0][:, 0] * wa0[h]).sum()
        if lo1 <= h < hi1:
            close(aux_out[h][1][:, 0], au[1].detach()[:, 0], what=f"aux1 {h}")
            ref_loss = ref_loss + (au[1][:, 0] * wa1[h]).sum()
    ref_loss.backward()
    if pins is not None:
        assert pins.calls == 3 * H and pins.disagree_far == 0, (pins.calls, pins.disagree_far)
    close(xg.grad, xc.grad, tol=2e-4, what="grad x")
    for h, lay in enumerate(layers):
        for k, prm in lay.named_parameters():
            want = sds[h][k].grad if sds[h][k].grad is not None else torch.zeros_like(sds[h][k])
            got = prm.grad if prm.grad is not None else torch.zeros_like(prm)
            close(got, want, tol=2e-4, what=f"grad head{h}.{k}")
