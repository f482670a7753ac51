"""disgat_gemm_planes (csrc/gemm_planes.hip): the f16x3 GEMM whose A operand arrives as two fp16 planes written by its
producer, against float64 - every epilogue form, ragged row counts, head-batched strided planes, plane output feeding
the next GEMM - and the plane-operand chain of a no-graph DISGAT forward (edge pass -> Z planes -> projection -> head
planes -> FuseLayer / DifHead classifier) against the fp32-operand chain it replaces."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_act(t, act):
    return {0: lambda v: v, 1: torch.nn.functional.elu, 2: lambda v: torch.nn.functional.leaky_relu(v, 0.01)}[act](t)


@pytest.mark.parametrize("m,hb,k,n,act,bias,init,planes_out", [
    (1000, 0, 256, 256, 0, False, False, False), (70001, 8, 256, 256, 1, False, False, True),
    (3333, 0, 2048, 256, 2, True, False, True), (513, 0, 64, 512, 0, False, True, False),
    (40000, 4, 128, 256, 1, True, True, True), (128, 0, 96, 256, 0, False, False, False),
    (127, 0, 64, 256, 2, True, True, True), (1, 0, 64, 256, 0, False, False, True), (100000, 0, 256, 2048, 0, False, False, False)])
def test_planes_gemm_vs_float64(m, hb, k, n, act, bias, init, planes_out):
    from edgedisentangle_ssl_amd import ops_gemm as og
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(m + k)
    if hb:
        z = torch.randn(m, hb, k, device=dev, generator=g) * torch.exp(torch.randn(m, 1, 1, device=dev, generator=g))
        a, w = z.permute(1, 0, 2), torch.randn(hb, k, n, device=dev, generator=g) * 0.1
    else:
        a = torch.randn(m, k, device=dev, generator=g) * torch.exp(torch.randn(m, 1, device=dev, generator=g))
        w = torch.randn(k, n, device=dev, generator=g) * 0.1
    H = max(hb, 1)
    b = torch.randn(H * n, device=dev, generator=g) if bias else None
    ini = torch.randn(m, H * n, device=dev, generator=g) if init else None
    ref = torch.bmm(a.double(), w.double()).permute(1, 0, 2).reshape(m, H * n) if hb else a.double() @ w.double()
    if b is not None:
        ref = ref + b.double()
    if ini is not None:
        ref = ref + ini.double()
    ref = _ref_act(ref, act)
    scale = float(ref.abs().max())
    ap = og.split_planes(a)
    assert float((ap.to_f32().double() - a.double()).abs().max()) <= 2.0 ** -22 * float(a.abs().max())   # the planes hold fp32
    bound = (ref.abs().max().float() * 1.01).reshape(1) if planes_out else None
    out, pl = og.linear_planes(ap, og.presplit_rm(w), n, b, ini, act, 0.01, True, bound)
    e_new = float((out.double() - ref).abs().max()) / scale
    e_old = float((og._forward(a, w, b, ini, act, 0.01).double() - ref).abs().max()) / scale
    assert e_new <= max(2.0 * e_old, 5e-7), (e_new, e_old)
    if pl is not None:
        assert float((pl.to_f32().double() - ref).abs().max()) / scale <= max(2.0 * e_old, 6e-7)
        only_pl = og.linear_planes(ap, og.presplit_rm(w), n, b, ini, act, 0.01, False, bound)
        assert only_pl[0] is None and torch.equal(only_pl[1].hi, pl.hi) and torch.equal(only_pl[1].lo, pl.lo)


def test_plane_output_feeds_the_next_gemm():
    """projection -> ELU -> planes -> fuser, the layer's dense chain: the second GEMM consumes what the first one's
    epilogue wrote, with the analytic bound (input bound x largest column abs-sum) as the hand-over scale."""
    from edgedisentangle_ssl_amd import ops_gemm as og
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    m, H, k, n = 5000, 8, 256, 256
    z = torch.randn(m, H, k, device=dev, generator=g)
    w1 = torch.randn(H, k, n, device=dev, generator=g) * 0.05
    w2 = torch.randn(H * n, 256, device=dev, generator=g) * 0.03
    b2 = torch.randn(256, device=dev, generator=g)
    zp = og.split_planes(z.permute(1, 0, 2))
    bound = torch.clamp(zp.bound * w1.abs().sum(1).max() * 1.001, min=1.0).reshape(1)
    _, hp = og.linear_planes(zp, og.presplit_rm(w1), n, None, None, og.ACT_ELU, 0.0, False, bound)
    out, _ = og.linear_planes(hp, og.presplit_rm(w2), 256, b2, None, og.ACT_LEAKY, 0.01)
    h64 = torch.nn.functional.elu(torch.bmm(z.permute(1, 0, 2).double(), w1.double())).permute(1, 0, 2).reshape(m, H * n)
    assert float(h64.abs().max()) <= float(bound)                       # the analytic bound dominates the true maximum
    ref = torch.nn.functional.leaky_relu(h64 @ w2.double() + b2.double(), 0.01)
    assert float((out.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


@pytest.mark.parametrize("gnn", ["AT", "SAGE", "GCN"])
def test_plane_chain_equals_fp32_chain_in_the_forward(gnn, monkeypatch):
    """The same no-graph forward with DISGAT_PLANES=1 (default) and =0: get_em, the aux scores, the DifHead loss.  With
    planes on, a layer whose fuser takes planes never writes the fp32 head buffer (HeadList stays empty)."""
    from types import SimpleNamespace
    from edgedisentangle_ssl_amd import DISGAT, layers, pretrainer, synth
    dev = torch.device("cuda")
    n, e, f, H = 20_000, 400_000, 256, 8
    a = SimpleNamespace(gnn_type=gnn, att=3, nhead=H, nhid=f, size=f, residue=False, residue_type=0, fuse_no_relu=False,
                        dropout=0.0, cls_layer=2, constrain_layer=0, sparse=True, model="DISGAT", dis_type=1, lr=0.01,
                        weight_decay=5e-4)
    torch.manual_seed(1)
    enc = DISGAT(a, nfeat=f, nhid=f, nclass=f, nheads=H, dropout=0.0).to(dev).eval()
    dif = pretrainer.DifHeadTrainer(a, enc, 1.0)
    for m_ in dif.models:
        m_.eval()
    fus = [dif.fuse1, dif.fuse2]
    graph = synth.powerlaw_graph(n, e, dev)
    x = synth.features(n, f, dev)
    pairs, _ = synth.uniform_pairs(n, 100_000, dev)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DISGAT_PLANES", flag)
        layers.clear_weight_cache(enc)
        for m_ in dif.models:
            layers.clear_weight_cache(m_)
        with torch.no_grad():
            em = enc.get_em(x, graph, fus)
            aux = enc.predict_adjs_sparse(x, graph, fus, [pairs])
            loss = dif.loss((x, graph))
            r = enc._run(x, graph, fus, heads_f32=False)
            edge_em = enc.get_edge_em(x, graph, fus)
        if flag == "1":
            assert r["heads"][0].planes is not None and r["heads"][0].fused is None and len(r["heads"][0]) == 0
        else:
            assert r["heads"][0].planes is None and r["heads"][0].fused is not None and len(r["heads"][0]) == H
        res[flag] = (em, [torch.cat([h[0] for h in layer], 1) for layer in aux], loss, [torch.stack(l) for l in edge_em])
    for l in range(2):
        ref = res["0"][0][l]
        assert float((res["1"][0][l] - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
        ref = res["0"][1][l]
        assert float((res["1"][1][l] - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
        ref = res["0"][3][l]                                   # get_edge_em always returns fp32 tensors
        assert float((res["1"][3][l] - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    assert abs(float(res["1"][2]) - float(res["0"][2])) <= 1e-5 * abs(float(res["0"][2]))


def test_foreign_fuser_still_receives_fp32_heads():
    """A fuser that is not this package's FuseLayer (e.g. the reference's own class, which torch.cat()s the list) gets
    a real list of fp32 tensors."""
    from types import SimpleNamespace
    from edgedisentangle_ssl_amd import DISGAT, synth
    dev = torch.device("cuda")
    n, f, H = 3000, 256, 8
    a = SimpleNamespace(gnn_type="AT", att=3, nhead=H, nhid=f, size=f, residue=False, residue_type=0, fuse_no_relu=False, dropout=0.0)
    torch.manual_seed(2)
    enc = DISGAT(a, nfeat=f, nhid=f, nclass=f, nheads=H, dropout=0.0).to(dev).eval()
    w = torch.randn(H * f, f, device=dev) * 0.02
    seen = []

    def foreign(feature_list, residue=None):
        seen.append((type(feature_list), len(feature_list)))
        return torch.cat(feature_list, dim=-1) @ w

    with torch.no_grad():
        out = enc.get_em(synth.features(n, f, dev), synth.powerlaw_graph(n, 40_000, dev), [foreign, foreign])
    assert all(ln == H for _t, ln in seen) and torch.isfinite(out[1]).all()
