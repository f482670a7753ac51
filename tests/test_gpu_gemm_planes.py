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


@pytest.mark.parametrize("m,hb,k,n_out,act,bias,init,bias2", [
    (5000, 8, 256, 8, 2, False, "shared", True), (70001, 8, 256, 8, 2, False, "shared", True), (1, 0, 64, 8, 2, True, None, False),
    (129, 4, 128, 16, 1, True, "full", True), (1000, 3, 256, 5, 0, False, None, True), (257, 8, 256, 1, 2, False, "shared", False),
    (12345, 2, 512, 12, 2, True, "shared", True)])
def test_planes_gemm_with_logits_epilogue_vs_float64(m, hb, k, n_out, act, bias, init, bias2):
    """disgat_gemm_planes_logits: L = act(A W1 + bias + init) W2^T + b2 without the hidden layer in memory - against float64
    and against the two-launch form (plane GEMM -> fp32 hidden -> skinny product) it replaces; ragged row counts (partial
    last tile), one row, odd head counts, every n_out class (vector and scalar stores), rows spanning 2^8 in magnitude."""
    from edgedisentangle_ssl_amd import ops_gemm as og
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(7 * m + k + n_out)
    H = max(hb, 1)
    z = torch.randn(m, H, k, device=dev, generator=g) * torch.exp2(4 * torch.rand(m, 1, 1, device=dev, generator=g) - 2)
    a = z.permute(1, 0, 2) if hb else z[:, 0]
    w1 = torch.randn(H, k, 256, device=dev, generator=g) * (1.0 / k ** 0.5)
    if not hb:
        w1 = w1[0]
    b1 = torch.randn(H * 256, device=dev, generator=g) * 0.3 if bias else None
    ini = {None: None, "shared": torch.randn(m, 256, device=dev, generator=g), "full": torch.randn(m, H * 256, device=dev, generator=g)}[init]
    lin2 = torch.nn.Linear(256, n_out, bias=bias2).to(dev)
    with torch.no_grad():
        lin2.weight.copy_(torch.randn(n_out, 256, device=dev, generator=g) * 0.2)
    hid = torch.bmm(a.double(), w1.double()).permute(1, 0, 2).reshape(m, H * 256) if hb else a.double() @ w1.double()
    if b1 is not None:
        hid = hid + b1.double()
    if ini is not None:
        hid = hid + (ini.double() if init == "full" else ini.double().repeat(1, H))
    hid = _ref_act(hid, act)
    ref = hid.reshape(m * H, 256) @ lin2.weight.detach().double().t() + (lin2.bias.detach().double() if bias2 else 0.0)
    ap = og.split_planes(a)
    w_rm = og.presplit_rm(w1)
    mid = (hid.abs().max().float() * 1.01).reshape(1)
    out = og.linear_planes_logits(ap, w_rm, b1, ini, act, 0.01, mid, og.presplit_logits(lin2.weight, lin2.bias))
    assert out.shape == (m * H, n_out)
    two = og.linear_planes(ap, w_rm, 256, b1, ini, act, 0.01)[0].view(m * H, 256)
    with torch.no_grad():
        two = og.skinny_linear(two, lin2) if og.skinny_ok(two, lin2) else lin2(two)
    # per row: the error relative to sum_k |hidden_k| |w_k| (what an fp32 dot product of that row is held to)
    scale = (hid.reshape(m * H, 256).abs() @ lin2.weight.detach().double().abs().t()).clamp_min(1e-30)
    e_new = float(((out.double() - ref).abs() / scale).max())
    e_two = float(((two.double() - ref).abs() / scale).max())
    assert e_new <= max(2.0 * e_two, 4e-7), (e_new, e_two)
    # a looser hand-over bound (the analytic one of the caller) costs nothing visible: fp16 hi + lo carry 22 bits
    out8 = og.linear_planes_logits(ap, w_rm, b1, ini, act, 0.01, mid * 8.0, og.presplit_logits(lin2.weight, lin2.bias))
    assert float(((out8.double() - ref).abs() / scale).max()) <= max(2.0 * e_two, 4e-7)


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


@pytest.mark.parametrize("gnn,att", [("AT", 3), ("SAGE", 3), ("GCN", 3), ("AT", 2), ("GCN", 1), ("SAGE", 2)])
def test_plane_chain_meets_a_reference_output(gnn, att, golden_dir):
    """the plane-operand chain (edge pass -> Z planes -> projection -> head planes -> fuser / DifHead
    classifier) never engages at the bundled graphs' nhid = 64, so it had only met the float64 oracle.  tiny256_*.npz holds
    the UNMODIFIED reference's outputs at a width the chain tiles (SURVEY 8(d)'s generator, N = 2 048, F_in = nhid = 256,
    H = 4; oracle/gen_golden.py --only tiny256): the five entry points and the three losses at north_star's 1e-4, with the
    launchers counted to prove the chain ran.  Round 5: GCN, att 2 - the reference's argparse default (utils.py:92) - with AT and
    SAGE, att 1 with GCN; for
    AT / GCN the projection + fuser pair of forward / get_em / the score entry points is the back-to-back launch
    (disgat_proj_fuse), DifHead keeps the plane GEMMs (its classifier reads the heads too)."""
    import os
    import numpy as np
    import inputs_common as ic
    from edgedisentangle_ssl_amd import _lib
    from test_gpu_parity import TOL, build, close
    from test_gpu_backward import _trainers
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, f"tiny256_{gnn}_att{att}.npz"))
    n, e, f, heads = 2048, 40960, 256, 4
    idx = ic.powerlaw_index(1234, n, e)
    ci = ic.coalesced_index_set(idx, n)
    adj = torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1]), (n, n)).to(dev)
    x = ic.features(71, n, f).to(dev)
    labels = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).integers(0, 4, n))
    pos, homo, het = ic.edge_sets(ci, labels, n)
    sup = [t.to(dev) for t in ic.sample_pairs(81, n, pos, "sup")]
    ho = [t.to(dev) for t in ic.sample_pairs(82, n, homo, "homo")]
    he = [t.to(dev) for t in ic.sample_pairs(83, n, het, "het")]
    a, enc, fus = build(gnn, att, heads, f, f, 400, dev)
    calls = {}
    real_call = _lib.call

    def counting(name, *args):
        calls[name] = calls.get(name, 0) + 1
        return real_call(name, *args)
    _lib.call = counting
    try:
        with torch.no_grad():
            fwd = enc(x, adj, fus)
            em = enc.get_em(x, adj, fus)
            n_planes = calls.get("disgat_gemm_planes", 0) + 2 * calls.get("disgat_proj_fuse", 0)
            assert (calls.get("disgat_proj_fuse", 0) >= 4) == (gnn in ("AT", "GCN")), calls
            adjs = enc.get_adjs(x, adj, fus)
            auxs = enc.predict_adjs_sparse(x, adj, fus, [sup[0]])
            eem = enc.get_edge_em(x, adj, fus)
            sup_t, dis_t, dif_t = _trainers(a, enc, 400, dev)
            l_sup = sup_t.loss((x, adj), sup[1], [sup[0]])
            l_dis = dis_t.loss((x, adj), [ho[1], he[1]], [ho[0], he[0]])
            before = calls.get("disgat_gemm_planes", 0) + calls.get("disgat_gemm_planes_logits", 0)
            l_dif = dif_t.loss((x, adj))
            n_dif = calls.get("disgat_gemm_planes", 0) + calls.get("disgat_gemm_planes_logits", 0) - before
    finally:
        _lib.call = real_call
    assert n_planes >= 2 * 4 and n_dif >= 6, calls           # forward + get_em: projection and fuser of both layers; DifHead: + classifier
    assert calls.get("disgat_gemm_planes_logits", 0) == 2, calls     # ... whose hidden layer goes straight into the logits (both layers)
    for key, t in (("forward", fwd), ("get_em_0", em[0]), ("get_em_1", em[1])):
        close(t[:256], g[key + "_head"], what=f"{key} head rows")
        scale = max(1.0, float(g[key + "_abssum"]) / t.numel() * 50)
        assert np.abs(t.double().sum(0).cpu().numpy() - g[key + "_colsum"]).max() <= TOL * scale * n, f"{key} column sums"
    stride, astride = max(1, ci.shape[1] // 2048), max(1, sup[0].shape[1] // 2048)
    for l in range(2):
        close(torch.stack([t[:, 0] for t in adjs[l]])[:, ::stride], g[f"adjs_{l}_sub"], what=f"adjs {l}")
        close(torch.stack([h[0][:, 0] for h in auxs[l]])[:, ::astride], g[f"aux_{l}_0_sub"], what=f"aux {l}")
        ee = torch.stack(list(eem[l]))
        close(ee[:, :64], g[f"edge_em_{l}_head"], what=f"edge_em {l} head rows")
        ref = g[f"edge_em_{l}_sum"]
        assert np.abs(ee.double().sum((1, 2)).cpu().numpy() - ref).max() <= TOL * max(1.0, np.abs(ref).max()) * 10
    for key, got in (("loss_sup", l_sup), ("loss_dis", l_dis), ("loss_dif", l_dif)):
        want = float(g[key])
        assert abs(got.item() - want) <= 1e-5 * max(1.0, abs(want)), (key, got.item(), want)
