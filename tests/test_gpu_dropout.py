"""Attention dropout inside the fused kernels (layers.py:394): the counter-based mask is
re-derived on the host (numpy splitmix64) and the result compared with a float64 CPU evaluation
of the reference formula under that same mask - forward and parameter gradients."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import inputs_common as ic
from test_gpu_parity import GNNS, close, dev, tiny_inputs  # noqa: F401

pytestmark = pytest.mark.gpu


def host_mask(seed, n_edges, H, p):
    k = np.arange(n_edges * H, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    keep = (z >> np.uint64(32)).astype(np.uint64) >= np.uint64(int(p * 4294967296.0))
    return torch.from_numpy(keep.reshape(n_edges, H).astype(np.float64) / (1.0 - p))


@pytest.mark.parametrize("chunk", [None, 8])
@pytest.mark.parametrize("gnn", GNNS)
@pytest.mark.parametrize("att", [1, 2, 3])
def test_dropout_forward_and_grads(dev, gnn, att, chunk, monkeypatch):
    import edgedisentangle_ssl_amd as pkg
    from edgedisentangle_ssl_amd import ops
    from oracle import disgat_oracle as orc
    if chunk is not None:
        monkeypatch.setattr(ops, "CHUNK", {1: chunk, 2: chunk, 3: chunk})
    p, H = 0.3, 4
    x, adj, n, _ = tiny_inputs(dev)
    idx, _, _ = ic.tiny_graph()
    ci = ic.coalesced_index_set(idx, n)
    layers = [ic.load_params(pkg.DisGALayer(16, 16, dropout=p, alpha=0.1, att_type=att, gnn_type=gnn), 300 + h).to(dev).train()
              for h in range(H)]
    torch.manual_seed(77)
    heads, e_list, _ = pkg.disga_heads(layers, x, adj)
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    mask = host_mask(seed, ci.shape[1], H, p)
    assert 0.55 < float((mask > 0).double().mean()) < 0.85
    wsum = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal((H, n, 16)))
    loss = sum((heads[h].double() * wsum[h].to(dev)).sum() for h in range(H))
    loss.backward()

    xc = x.cpu().double()
    r, c = ci[0], ci[1]
    ref_loss = 0.0
    for h, lay in enumerate(layers):
        sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in lay.state_dict().items()}
        e = orc.pair_score(att, xc, sd["W"], sd["a"], r, c)
        a_d = orc.sp_softmax(r, torch.sigmoid(e), n) * mask[:, h:h + 1]
        if gnn == "AT":
            hp = orc.sp_matmul(r, c, a_d, xc @ sd["W_em"])
        elif gnn == "SAGE":
            rowsum = torch.zeros(n, 1, dtype=torch.float64).index_add_(0, r, a_d.detach())
            hp = torch.cat([xc, orc.sp_matmul(r, c, a_d, xc) / (rowsum + 1)], -1) @ sd["ag_layer.proj.weight"].t()
        else:
            hp = orc.sp_matmul(r, c, a_d, xc @ sd["ag_layer.weight"]) + sd["ag_layer.bias"]
        out = F.elu(hp)
        close(heads[h], out.detach(), what=f"head {h}")
        close(e_list[h][:, 0], e.detach()[:, 0], what=f"edge_e {h}")
        lh = (out * wsum[h]).sum()
        lh.backward()
        for k, prm in lay.named_parameters():
            want = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])     # att 2 never uses `a`
            got = prm.grad if prm.grad is not None else torch.zeros_like(prm)
            close(got, want, tol=2e-4, what=f"grad head{h}.{k}")


def test_dropout_off_in_eval_and_seed_repeatable(dev):
    import edgedisentangle_ssl_amd as pkg
    x, adj, n, _ = tiny_inputs(dev)
    layers = [ic.load_params(pkg.DisGALayer(16, 16, dropout=0.5, alpha=0.1, att_type=3, gnn_type="AT"), 400 + h).to(dev)
              for h in range(4)]
    for l in layers:
        l.eval()
    with torch.no_grad():
        a = pkg.disga_heads(layers, x, adj)[0]
        b = pkg.disga_heads(layers, x, adj)[0]
        assert all(torch.equal(u, v) for u, v in zip(a, b))
        for l in layers:
            l.train()
        torch.manual_seed(1)
        c = pkg.disga_heads(layers, x, adj)[0]
        torch.manual_seed(1)
        d = pkg.disga_heads(layers, x, adj)[0]
        e = pkg.disga_heads(layers, x, adj)[0]
    assert all(torch.equal(u, v) for u, v in zip(c, d))
    assert not all(torch.equal(u, v) for u, v in zip(d, e))
    assert not all(torch.equal(u, v) for u, v in zip(a, c))
