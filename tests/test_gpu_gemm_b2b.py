"""disgat_proj_fuse (csrc/gemm_b2b.hip): the per-head projection + ELU + FuseLayer as one back-to-back GEMM, against
float64 on every instantiation, with and without the GCN bias, ragged row counts, a fuser without activation - and against
the two-launch plane chain it replaces (/root/reference/layers.py:397-399, 404-407, 508, 896-921)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(m, H, k1, n1, n2, bias1, bias2, seed):
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(seed)
    z = torch.randn(m, H, k1, device=dev, generator=g) * torch.exp(0.5 * torch.randn(m, 1, 1, device=dev, generator=g))
    w1 = torch.randn(H, k1, n1, device=dev, generator=g) * (1.5 / k1 ** 0.5)
    # asymmetric, head-dependent fuser weight: a row <-> column or head mix-up cannot cancel
    w2 = torch.randn(H * n1, n2, device=dev, generator=g) * (1.0 / (H * n1) ** 0.5) + 0.01 * torch.linspace(-1, 1, n2, device=dev)
    b1 = torch.randn(H * n1, device=dev, generator=g) * 0.3 if bias1 else None
    b2 = torch.randn(n2, device=dev, generator=g) if bias2 else None
    return z, w1, w2, b1, b2


def _ref(z, w1, w2, b1, b2, act2):
    m, H, _ = z.shape
    mid = torch.bmm(z.permute(1, 0, 2).double(), w1.double())              # [H, M, N1]
    if b1 is not None:
        mid = mid + b1.double().view(H, 1, -1)
    mid = torch.nn.functional.elu(mid).permute(1, 0, 2).reshape(m, -1)
    out = mid @ w2.double()
    if b2 is not None:
        out = out + b2.double()
    return (torch.nn.functional.leaky_relu(out, 0.01) if act2 else out), mid


@pytest.mark.parametrize("m,H,k1,n1,n2,bias1,bias2,act2", [
    (1000, 8, 256, 256, 256, False, True, 2),        # the benchmarked instantiation, ragged last block
    (128, 2, 256, 256, 256, True, True, 2),          # GCN bias
    (4097, 4, 128, 128, 128, False, False, 0),       # fuse_no_relu, no fuser bias
    (515, 8, 64, 64, 64, True, True, 2),             # nhid = 64 (the bundled graphs' width)
    (300, 3, 128, 96, 256, False, True, 2),          # odd head count, N1 not a power of two
    (1, 2, 256, 64, 128, True, False, 2),            # one row, two chunks per head
    (20000, 16, 64, 64, 64, False, True, 2),
    (70001, 8, 256, 256, 256, False, True, 2)])
def test_proj_fuse_vs_float64(m, H, k1, n1, n2, bias1, bias2, act2):
    from edgedisentangle_ssl_amd import ops_gemm as og
    z, w1, w2, b1, b2 = _case(m, H, k1, n1, n2, bias1, bias2, m + n1)
    ref, mid = _ref(z, w1, w2, b1, b2, act2)
    zp = og.split_planes(z.permute(1, 0, 2))
    # the analytic bound the layer uses: |elu(v)| <= max(|v|, 1), |v| <= bound(Z) x largest column abs-sum (+ max |bias|)
    pre = zp.bound * w1.abs().sum(1).max() + (b1.abs().max() if b1 is not None else 0.0)
    bound = torch.clamp(pre * 1.001, min=1.0).reshape(1)
    assert float(mid.abs().max()) <= float(bound)
    out = og.proj_fuse(zp, og.presplit_b2b(w1, w2), b1, b2, bound, n1, n2, act2, 0.01)
    scale = float(ref.abs().max())
    err = float((out.double() - ref).abs().max()) / scale
    # yardstick: the same contraction on the fp32-operand kernels (two launches through an fp32 head buffer)
    h32 = og._forward(z.permute(1, 0, 2), w1, b1, None, og.ACT_ELU, 0.0)
    two = og._forward(h32, w2, b2, None, og.ACT_LEAKY if act2 else og.ACT_NONE, 0.01) if (H * n1) % 32 == 0 and n2 % 32 == 0 else None
    e_two = float((two.double() - ref).abs().max()) / scale if two is not None else 0.0
    assert err <= max(2.0 * e_two, 1e-6), (err, e_two)


def test_proj_fuse_tracks_a_wide_dynamic_range():
    """Rows spanning 2^20 in magnitude: every output row is held to ITS OWN scale (the f16x3 operands carry 2^-23 per
    element down to 2^-27 of the operand maximum)."""
    from edgedisentangle_ssl_amd import ops_gemm as og
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(11)
    m, H, k1, n1, n2 = 2048, 4, 256, 256, 256
    z = torch.randn(m, H, k1, device=dev, generator=g) * (2.0 ** -torch.randint(0, 14, (m, 1, 1), device=dev, generator=g).float())
    w1 = torch.randn(H, k1, n1, device=dev, generator=g) * 0.1
    w2 = torch.randn(H * n1, n2, device=dev, generator=g) * 0.03
    ref, mid = _ref(z, w1, w2, None, None, 0)
    zp = og.split_planes(z.permute(1, 0, 2))
    bound = torch.clamp(zp.bound * w1.abs().sum(1).max() * 1.001, min=1.0).reshape(1)
    out = og.proj_fuse(zp, og.presplit_b2b(w1, w2), None, None, bound, n1, n2, 0, 0.0)
    mag = mid.abs() @ w2.double().abs()                       # each element's own rounding scale
    err = float(((out.double() - ref).abs() / mag).max())
    # yardstick: the two-launch plane chain in the same metric (its ELU is the same exp(v) - 1, whose absolute error of one
    # ulp of 1.0 is what shows on the rows that are 2^-13 of the largest)
    _, hp = og.linear_planes(zp, og.presplit_rm(w1), n1, None, None, og.ACT_ELU, 0.0, False, bound)
    two = og.linear_planes(hp, og.presplit_rm(w2), n2, None, None, og.ACT_NONE, 0.0)[0]
    e_two = float(((two.double() - ref).abs() / mag).max())
    assert err <= max(1.5 * e_two, 3e-6), (err, e_two)


@pytest.mark.parametrize("gnn", ["AT", "GCN"])
def test_b2b_forward_equals_two_launch_chain(gnn, monkeypatch):
    """The same no-graph forwards with DISGAT_B2B=1 (default) and =0 (projection and fuser as two plane GEMMs): get_em, the
    aux scores of predict_adjs_sparse; with it on, no head buffer exists in any form and the launcher is counted."""
    from types import SimpleNamespace
    from edgedisentangle_ssl_amd import DISGAT, _lib, layers, pretrainer, synth
    dev = torch.device("cuda")
    a = SimpleNamespace(gnn_type=gnn, att=3, nhead=4, nhid=256, size=256, residue=False, residue_type=0, fuse_no_relu=False,
                        dropout=0.0, cls_layer=2, constrain_layer=0, sparse=True, model="DISGAT", dis_type=1, lr=0.01,
                        weight_decay=5e-4)
    torch.manual_seed(5)
    n = 3000
    graph = synth.powerlaw_graph(n, 20 * n, dev)
    x = synth.features(n, 256, dev)
    enc = DISGAT(a, nfeat=256, nhid=256, nclass=256, nheads=4, dropout=0.0).to(dev).eval()
    sup = pretrainer.SupEdgeTrainer(a, enc, 1.0)
    dif = pretrainer.DifHeadTrainer(a, enc, 1.0)
    for m_ in sup.models + dif.models:
        m_.eval()
    (si, sl), _, _ = synth.ssl_lists(graph, synth.node_labels(n, dev))
    calls = []
    real = _lib.call

    def spy(name, *args):
        calls.append(name)
        return real(name, *args)
    monkeypatch.setattr(_lib, "call", spy)

    def run():
        layers.clear_weight_cache(enc)
        layers.clear_weight_cache(sup.fuse1)
        layers.clear_weight_cache(sup.fuse2)
        with torch.no_grad():
            em = enc.get_em(x, graph, [sup.fuse1, sup.fuse2])
            aux = enc.predict_adjs_sparse(x, graph, [sup.fuse1, sup.fuse2], [si])
            # DifHead's classifier reads the heads too: its pass keeps the head planes (no deferral), whatever the switch says
            n0 = len(calls)
            dl = dif.loss((x, graph))
            assert "disgat_proj_fuse" not in calls[n0:]
        return em, aux, float(dl)
    monkeypatch.setenv("DISGAT_B2B", "1")
    em1, aux1, dl1 = run()
    n_b2b = calls.count("disgat_proj_fuse")
    assert n_b2b == 4, calls
    calls.clear()
    monkeypatch.setenv("DISGAT_B2B", "0")
    em0, aux0, dl0 = run()
    assert "disgat_proj_fuse" not in calls and dl1 == dl0
    for l in range(2):
        sc = float(em0[l].abs().max())
        assert float((em1[l] - em0[l]).abs().max()) <= 2e-6 * max(1.0, sc)
    for l in range(2):
        for h in range(4):
            sc = float(aux0[l][h][0].abs().max())
            assert float((aux1[l][h][0] - aux0[l][h][0]).abs().max()) <= 1e-5 * max(1.0, sc)
