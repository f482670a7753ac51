"""The split-precision MFMA GEMMs (f16x3: two fp16 planes, 3 products - the default; split6: three bf16 planes, 6
products) against float64, with hipBLASLt's fp32 GEMM as the accuracy yardstick."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _err(x, ref):
    return float((x.double() - ref).abs().max() / ref.abs().max())


@pytest.fixture(params=["f16x3", "split6"])
def scheme(request, monkeypatch):
    monkeypatch.setenv("DISGAT_GEMM", request.param)
    return request.param


@pytest.mark.parametrize("m,k,n", [(1000, 256, 256), (4096, 64, 512), (777, 2048, 256), (130, 32, 128),
                                   (70001, 256, 256), (33000, 96, 2048), (19793, 64, 64), (2277, 128, 96), (300, 256, 32),
                                   (255, 64, 512), (257, 256, 2048)])
def test_plain_accuracy_matches_fp32_gemm(m, k, n, scheme):
    from edgedisentangle_ssl_amd import ops_gemm
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(m, k, device="cuda", generator=g) * torch.exp(torch.randn(m, 1, device="cuda", generator=g))
    w = torch.randn(k, n, device="cuda", generator=g) * 0.1
    ref = a.double() @ w.double()
    got = ops_gemm.linear(a, w)
    blas = a @ w
    e_got, e_blas = _err(got, ref), _err(blas, ref)
    assert e_got <= max(2.0 * e_blas, 3e-7), (e_got, e_blas)


def test_wide_dynamic_range_keeps_per_element_precision(scheme):
    """One global power-of-two scale per operand (f16x3): an outlier 2^12 times the typical magnitude and rows 2^-10
    below it (2^-22 of the maximum; the scheme keeps full per-element precision down to 2^-27 of it) must not cost
    the ordinary elements their precision - every row is checked against its own magnitude."""
    from edgedisentangle_ssl_amd import ops_gemm
    g = torch.Generator(device="cuda").manual_seed(7)
    m, k, n = 3000, 256, 256
    a = torch.randn(m, k, device="cuda", generator=g)
    a[5, 17] = 2.0 ** 12
    a[1000:1100] *= 2.0 ** -10
    w = torch.randn(k, n, device="cuda", generator=g) * 0.1
    w[3, 9] = 300.0
    ref = a.double() @ w.double()
    row_scale = (a.double().abs() @ w.double().abs()).clamp_min(1e-300)       # sum_k |a||w|: the natural error scale
    e_got = float(((ops_gemm.linear(a, w).double() - ref).abs() / row_scale).max())
    e_blas = float((((a @ w).double() - ref).abs() / row_scale).max())
    assert e_got <= max(2.0 * e_blas, 2e-7), (e_got, e_blas)


def test_zero_and_tiny_operands(scheme):
    from edgedisentangle_ssl_amd import ops_gemm
    a = torch.zeros(300, 64, device="cuda")
    w = torch.randn(64, 128, device="cuda")
    assert torch.equal(ops_gemm.linear(a, w), torch.zeros(300, 128, device="cuda"))
    a = torch.randn(300, 64, device="cuda") * 1e-25
    ref = a.double() @ w.double()
    assert _err(ops_gemm.linear(a, w), ref) < 1e-6


def test_batched_heads_bias_init_activation_and_grads():
    from edgedisentangle_ssl_amd import ops_gemm
    g = torch.Generator(device="cuda").manual_seed(2)
    H, m, k, n = 8, 1500, 64, 128
    z = torch.randn(m, H, k, device="cuda", generator=g)
    zt = z.permute(1, 0, 2)                                   # strided [H,M,K] view, as in disga_heads
    w = (torch.randn(H, k, n, device="cuda", generator=g) * 0.2).requires_grad_(True)
    bias = torch.randn(H * n, device="cuda", generator=g).requires_grad_(True)
    init = torch.randn(m, H * n, device="cuda", generator=g).requires_grad_(True)
    zt_g = zt.detach().clone().requires_grad_(True)
    for act, slope, fn in ((ops_gemm.ACT_ELU, 0.0, torch.nn.functional.elu),
                           (ops_gemm.ACT_LEAKY, 0.1, lambda t: torch.nn.functional.leaky_relu(t, 0.1)),
                           (ops_gemm.ACT_NONE, 0.0, lambda t: t)):
        out = ops_gemm.linear(zt_g, w, bias, init, act, slope)
        ref = fn((torch.bmm(zt.double(), w.double()) + bias.double().view(H, 1, n)).permute(1, 0, 2).reshape(m, H * n)
                 + init.double())
        assert _err(out, ref.detach()) < 1e-6
        wsum = torch.randn(m, H * n, device="cuda", generator=g)
        grads = torch.autograd.grad((out * wsum).sum(), [zt_g, w, bias, init])
        zr = zt.detach().double().requires_grad_(True)
        wr, br, ir = (t.detach().double().requires_grad_(True) for t in (w, bias, init))
        refo = fn((torch.bmm(zr, wr) + br.view(H, 1, n)).permute(1, 0, 2).reshape(m, H * n) + ir)
        rg = torch.autograd.grad((refo * wsum.double()).sum(), [zr, wr, br, ir])
        for a_, b_ in zip(grads, rg):
            assert _err(a_, b_) < 1e-5


def test_large_batched_with_epilogue():
    """Many tiles, ragged M, strided batched A, every epilogue term."""
    from edgedisentangle_ssl_amd import ops_gemm
    g = torch.Generator(device="cuda").manual_seed(3)
    H, m, k, n = 8, 12345, 64, 128                            # 97 M-tiles x 8 heads = 776 tiles
    z = torch.randn(m, H, k, device="cuda", generator=g)
    zt = z.permute(1, 0, 2)
    w = torch.randn(H, k, n, device="cuda", generator=g) * 0.2
    bias = torch.randn(H * n, device="cuda", generator=g)
    init = torch.randn(m, H * n, device="cuda", generator=g)
    out = ops_gemm.linear(zt, w, bias, init, ops_gemm.ACT_ELU)
    ref = torch.nn.functional.elu((torch.bmm(zt.double(), w.double()) + bias.double().view(H, 1, n)).permute(1, 0, 2)
                                  .reshape(m, H * n) + init.double())
    assert _err(out, ref) < 1e-6
    assert torch.equal(out, ops_gemm.linear(zt, w, bias, init, ops_gemm.ACT_ELU))      # deterministic


def test_fallback_shapes_use_blas():
    """What still goes to hipBLASLt: small products off the kernels' tiling (launch-bound: one library launch is the cheaper
    route)."""
    from edgedisentangle_ssl_amd import _lib, ops_gemm
    a = torch.randn(50, 16, device="cuda")
    w = torch.randn(16, 24, device="cuda")
    calls = []
    real = _lib.call
    _lib.call = lambda name, *args: (calls.append(name), real(name, *args))[1]
    try:
        got = ops_gemm.linear(a, w, act=ops_gemm.ACT_LEAKY, slope=0.01)
    finally:
        _lib.call = real
    assert not any(c.startswith("disgat_gemm") for c in calls)
    assert torch.allclose(got, torch.nn.functional.leaky_relu(a @ w), atol=1e-5)


@pytest.mark.parametrize("m,k,n,bias", [(20000, 1433, 512, False), (19793, 8710, 512, True), (40000, 250, 70, True)])
def test_large_off_tile_shapes_are_padded_onto_the_kernels(m, k, n, bias):
    """Raw bag-of-words widths (Cora 1 433, cora_full 8 710: /root/reference/main.py:108 takes the width from the data
    file) and narrow outputs: K zero-padded to 32, N to the column granule, on the f16x3 kernels - no library GEMM - with the
    padded copy of the constant operand reused; result against float64, hipBLASLt's fp32 GEMM as yardstick; gradients of the
    differentiable form equal the library's."""
    from edgedisentangle_ssl_amd import _lib, ops_gemm
    g = torch.Generator(device="cuda").manual_seed(k)
    a = torch.randn(m, k, device="cuda", generator=g)
    w = (torch.randn(k, n, device="cuda", generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(n, device="cuda", generator=g) if bias else None
    calls = []
    real = _lib.call
    _lib.call = lambda name, *args: (calls.append(name), real(name, *args))[1]
    try:
        out = ops_gemm.linear(a, w, b, None, ops_gemm.ACT_LEAKY, 0.01)
        n_pad = len(ops_gemm._KPAD)
        out2 = ops_gemm.linear(a, w, b, None, ops_gemm.ACT_LEAKY, 0.01)
    finally:
        _lib.call = real
    assert calls.count("disgat_gemm_f16x3") == 2 and len(ops_gemm._KPAD) == n_pad           # second call reused the padded copy
    assert out.shape == (m, n) and out.is_contiguous() and torch.equal(out, out2)
    ref = a.double() @ w.detach().double() + (b.double() if bias else 0.0)
    ref = torch.nn.functional.leaky_relu(ref, 0.01)
    blas = torch.nn.functional.leaky_relu(a @ w.detach() + (b if bias else 0.0), 0.01)
    scale = float(ref.abs().max())
    e_ours, e_blas = float((out.double() - ref).abs().max()) / scale, float((blas.double() - ref).abs().max()) / scale
    assert e_ours <= max(2 * e_blas, 1e-6), (e_ours, e_blas)
    # (gradient on the form without activation: of 1e7 outputs a few sit within rounding of the leaky-ReLU kink, where two
    # fp32 evaluations legitimately take different sides)
    co = torch.randn(m, n, device="cuda", generator=g)
    (gw,) = torch.autograd.grad((ops_gemm.linear(a, w, b) * co).sum(), [w])
    w2 = w.detach().clone().requires_grad_(True)
    (gw2,) = torch.autograd.grad(((a @ w2 + (b if bias else 0.0)) * co).sum(), [w2])
    assert float((gw - gw2).abs().max()) <= 2e-4 * float(gw2.abs().max())
    ops_gemm._KPAD.clear()


@pytest.mark.parametrize("gnn", ["AT", "SAGE", "GCN"])
@pytest.mark.parametrize("train", [False, True])
def test_analytic_operand_bounds_hold(gnn, train):
    """disga_heads hands the f16x3 GEMMs analytic upper bounds of max |Z| and max |fused| instead of measuring the two
    largest operands; a bound below the true maximum would overflow fp16."""
    import inputs_common as ic
    import edgedisentangle_ssl_amd as pkg
    from edgedisentangle_ssl_amd import ops_gemm
    dev = torch.device("cuda")
    idx, vals, n = ic.tiny_graph()
    adj = torch.sparse_coo_tensor(idx, vals, (n, n)).to(dev)
    x = (ic.features(5, n, 64) * 30.0).to(dev)
    layers = [ic.load_params(pkg.DisGALayer(64, 128, dropout=0.5, alpha=0.1, att_type=3, gnn_type=gnn), 40 + h).to(dev)
              for h in range(4)]
    for lay in layers:
        lay.train(train)
    seen = []
    real = ops_gemm._forward

    def spy(a, w, bias, init, act, slope, a_amax=None, w_split=None):
        if a_amax is not None:
            seen.append((float(a.abs().max()), float(a_amax)))
        return real(a, w, bias, init, act, slope, a_amax, w_split)

    ops_gemm._forward = spy
    try:
        torch.manual_seed(0)
        heads, _, _ = pkg.disga_heads(layers, x, adj)
    finally:
        ops_gemm._forward = real
    assert len(seen) >= 3
    for true_max, bound in seen:
        assert bound >= true_max, (true_max, bound)
    assert float(heads.fused_amax) >= float(heads.fused.abs().max())
    assert torch.isfinite(heads.fused).all()


def test_weight_split_reconstructs_strided_and_expanded_weights():
    """disgat_split_f16 on a transposed view, a column slice and a stride-0 head expansion: hi + lo * 2^-11 must
    reproduce w * s to 2^-22 of each element, s a power of two placing max |w| in [2^13, 2^14)."""
    from edgedisentangle_ssl_amd import ops_gemm
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.randn(300, 200, device="cuda", generator=g) * 3.0
    big = torch.randn(700, 520, device="cuda", generator=g) * 0.02     # > 131072 elements: the two-launch form
    cases = [base, base.t(), base[:, 40:168], base[::2, :].t().unsqueeze(0).expand(4, 200, 150), big, big.t()[:, 100:]]
    for w in cases:
        planes, s = ops_gemm.split_weight_f16(w)
        w3 = w if w.dim() == 3 else w.unsqueeze(0)
        p = planes.view(torch.float16).float()                       # [hb, 2, N, K]
        # the one-launch form (small weights) and the two-launch form store the same planes: exactly these
        t = (w3 * s).transpose(1, 2)
        h = t.half()
        assert torch.equal(planes.view(torch.float16)[:, 0], h)
        assert torch.equal(planes.view(torch.float16)[:, 1], ((t - h.float()) * 2048.0).half())
        rec = (p[:, 0] + p[:, 1] / 2048.0).transpose(1, 2)            # [hb, K, N]
        ref = w3 * s
        assert float(s) == 2.0 ** round(float(torch.log2(s)))        # a power of two
        assert 2.0 ** 13 <= float((w3.abs().max() * s)) < 2.0 ** 14
        assert float(((rec - ref).abs() / ref.abs().clamp_min(1e-3)).max()) < 2.0 ** -21


def test_amax_batched_strided_view():
    from edgedisentangle_ssl_amd import ops_gemm
    z = torch.randn(777, 8, 64, device="cuda")
    z[123, 5, 17] = -99.5
    zt = z.permute(1, 0, 2)                                           # [H, M, K] strided, as the projection GEMM sees Z
    assert float(ops_gemm.amax(zt)) == 99.5
    assert float(ops_gemm.amax(z.view(777, 512))) == 99.5
    assert float(ops_gemm.amax(torch.zeros(10, 64, device="cuda"))) == 0.0


@pytest.mark.parametrize("m,k,n", [(20000, 256, 384), (70001, 128, 2048), (9000, 2048, 128)])
def test_weight_gradient_split_k_kernel(m, k, n):
    """a^T g over the row dimension (gemm_f16x3_tn_kernel: transposing LDS reads, split-K partials) against float64,
    hipBLASLt's fp32 as yardstick; ragged M (not a multiple of 32 or of the split size)."""
    from edgedisentangle_ssl_amd import ops_gemm
    gen = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randn(m, k, device="cuda", generator=gen) * torch.exp(torch.randn(m, 1, device="cuda", generator=gen))
    g = torch.randn(m, n, device="cuda", generator=gen) * 0.3
    assert ops_gemm._tn_ok(a, g, k, n)
    got = ops_gemm._weight_grad(a, g, ops_gemm.amax(a), ops_gemm.amax(g))
    ref = a.double().t() @ g.double()
    e_got, e_blas = _err(got, ref), _err(a.t() @ g, ref)
    assert got.shape == (k, n)
    assert e_got <= max(2.0 * e_blas, 3e-7), (e_got, e_blas)
    assert torch.equal(got, ops_gemm._weight_grad(a, g, ops_gemm.amax(a), ops_gemm.amax(g)))      # deterministic


def test_weight_gradient_batched_strided_through_autograd():
    from edgedisentangle_ssl_amd import ops_gemm
    gen = torch.Generator(device="cuda").manual_seed(12)
    H, m, k, n = 4, 10000, 128, 256
    z = torch.randn(m, H, k, device="cuda", generator=gen)
    zt = z.permute(1, 0, 2)                                   # strided [H,M,K] view, as in disga_heads
    w = (torch.randn(H, k, n, device="cuda", generator=gen) * 0.2).requires_grad_(True)
    out = ops_gemm.linear(zt, w, None, None, ops_gemm.ACT_ELU)
    wsum = torch.randn(m, H * n, device="cuda", generator=gen)
    (gw,) = torch.autograd.grad((out * wsum).sum(), [w])
    wr = w.detach().double().requires_grad_(True)
    refo = torch.nn.functional.elu(torch.bmm(zt.double(), wr).permute(1, 0, 2).reshape(m, H * n))
    (rg,) = torch.autograd.grad((refo * wsum.double()).sum(), [wr])
    assert _err(gw, rg) < 2e-6


@pytest.mark.parametrize("m,k,n,bias", [(1000, 256, 8, True), (4097, 512, 4, False), (333, 256, 16, True), (5, 256, 3, True)])
def test_skinny_linear_matches_fp64(m, k, n, bias):
    """disgat_linear_skinny (the DifHead classifier's hidden -> nhead layer, models.py:523-543): fp32 FMA result against a
    float64 product, forward and the gradients of its autograd wrapper."""
    from edgedisentangle_ssl_amd import ops_gemm
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(m + n)
    lin = torch.nn.Linear(k, n, bias=bias).to(dev)
    xbuf = torch.randn(m, k + 4, generator=g).to(dev)                 # a row stride that is not K
    x = xbuf[:, :k].requires_grad_(True)
    assert ops_gemm.skinny_ok(x, lin)
    y = ops_gemm.skinny_linear(x, lin)
    ref = x.detach().double() @ lin.weight.detach().double().t() + (lin.bias.detach().double() if bias else 0.0)
    assert float((y.detach().double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    w = torch.randn(m, n, generator=g).to(dev)
    (y * w).sum().backward()
    assert torch.allclose(x.grad, w @ lin.weight.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(lin.weight.grad, w.t() @ x.detach(), rtol=1e-4, atol=1e-4)
    if bias:
        assert torch.allclose(lin.bias.grad, w.sum(0), rtol=1e-5, atol=1e-4)
    assert not ops_gemm.skinny_ok(x[:, :128], lin) and not ops_gemm.skinny_ok(x.cpu(), lin)


@pytest.mark.parametrize("m,hb,k,n,act", [(2708, 8, 64, 64, 1), (19793, 8, 64, 64, 1), (1000, 4, 128, 32, 2), (513, 2, 256, 96, 0),
                                          # few row blocks: the launcher shares a row block's column chunks among 2-16 workgroups
                                          (2277, 1, 64, 1024, 0), (2708, 1, 64, 1024, 2), (700, 2, 128, 512, 1), (5000, 1, 256, 384, 0),
                                          (40000, 1, 64, 256, 1), (1, 1, 64, 128, 0)])
def test_narrow_outputs_run_on_the_library_kernel(m, hb, k, n, act, monkeypatch):
    """N % 32 (not 128) for K = 64 / 128 / 256: the register-stationary kernel (csrc/gemm_rs.hip) takes the nhid = 64 layers
    of the bundled graphs (per-head projection [64 -> 64] x 8 heads, batched, with bias / additive input / activation) that
    used to fall to hipBLASLt; against float64, forward and both gradients, and the launcher counted."""
    from edgedisentangle_ssl_amd import _lib, ops_gemm
    g = torch.Generator(device="cuda").manual_seed(m + n)
    z = torch.randn(m, hb, k, device="cuda", generator=g)
    a = z.permute(1, 0, 2).requires_grad_(True)
    w = (torch.randn(hb, k, n, device="cuda", generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(hb * n, device="cuda", generator=g)
    ini = torch.randn(m, hb * n, device="cuda", generator=g)
    calls = []
    real = _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *args: (calls.append(name), real(name, *args))[1])
    out = ops_gemm.linear(a, w, b, ini, act, 0.01)
    assert calls.count("disgat_gemm_f16x3") == 1
    pre = (torch.bmm(a.detach().double(), w.detach().double()).permute(1, 0, 2).reshape(m, hb * n) + b.double() + ini.double())
    ref = {0: lambda t: t, 1: torch.nn.functional.elu, 2: lambda t: torch.nn.functional.leaky_relu(t, 0.01)}[act](pre)
    assert _err(out.detach(), ref) < 1e-6
    go = torch.randn(m, hb * n, device="cuda", generator=g)
    out.backward(go)
    ad = a.detach().double().requires_grad_(True)
    wd = w.detach().double().requires_grad_(True)
    pre = torch.bmm(ad, wd).permute(1, 0, 2).reshape(m, hb * n) + b.double() + ini.double()
    {0: lambda t: t, 1: torch.nn.functional.elu, 2: lambda t: torch.nn.functional.leaky_relu(t, 0.01)}[act](pre).backward(go.double())
    assert _err(a.grad, ad.grad) < 2e-6 and _err(w.grad, wd.grad) < 2e-5


@pytest.mark.parametrize("m,hb,k,n", [(2708, 1, 64, 512), (2277, 8, 64, 64), (19793, 1, 64, 512), (19793, 8, 64, 64), (600, 1, 128, 96),
                                      (513, 1, 4, 4), (70001, 2, 36, 260), (512, 1, 256, 1024)])
def test_small_result_weight_gradients(m, hb, k, n, monkeypatch):
    """disgat_wgrad_small (csrc/wgrad_small.hip): a^T g with a few output tiles and a long row reduction - the nhid = 64 weight
    gradients of the bundled graphs - against float64, hipBLASLt's fp32 as yardstick; ragged tiles (K, N not multiples of 64),
    ragged row ranges, head-batched strided views; deterministic; and _wgrad_blas really dispatches to it."""
    from edgedisentangle_ssl_amd import _lib, ops_gemm
    gen = torch.Generator(device="cuda").manual_seed(m + k + n)
    if hb > 1:
        a = (torch.randn(m, hb, k, device="cuda", generator=gen) * 2.0).permute(1, 0, 2)        # [H, M, K] view of [M, H, K]
        g = (torch.randn(m, hb, n, device="cuda", generator=gen) * 0.3).permute(1, 0, 2)
        ref = torch.bmm(a.double().transpose(1, 2), g.double())
        blas = torch.bmm(a.transpose(1, 2), g)
    else:
        a = torch.randn(m, k, device="cuda", generator=gen) * 2.0
        g = torch.randn(m, n, device="cuda", generator=gen) * 0.3
        ref = a.double().t() @ g.double()
        blas = a.t() @ g
    calls = []
    real = _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *args: (calls.append(name), real(name, *args))[1])
    got = ops_gemm._wgrad_blas(a, g)
    assert calls == ["disgat_wgrad_small"]
    assert got.shape == ref.shape
    e_got, e_blas = _err(got, ref), _err(blas, ref)
    assert e_got <= max(2.0 * e_blas, 3e-7), (e_got, e_blas)
    assert torch.equal(got, ops_gemm._wgrad_blas(a, g))
    monkeypatch.setenv("DISGAT_WGRAD_SMALL", "0")
    calls.clear()
    other = ops_gemm._wgrad_blas(a, g)
    assert "disgat_wgrad_small" not in calls and _err(other, ref) < 1e-5


def test_weight_bound_matches_the_aten_chain():
    """disgat_weight_bound: largest column abs-sum of a (batched, strided) weight and max(floor, in_bound * it * scale)."""
    from edgedisentangle_ssl_amd import ops_gemm
    g = torch.Generator(device="cuda").manual_seed(3)
    bound = torch.tensor([2.5], device="cuda")
    for w in (torch.randn(8, 64, 64, device="cuda", generator=g), torch.randn(300, 70, device="cuda", generator=g).t(),
              torch.randn(5, 33, 17, device="cuda", generator=g).transpose(1, 2), torch.randn(1, 1, device="cuda", generator=g),
              torch.randn(600, 400, device="cuda", generator=g)):            # the last one: above the one-block limit (ATen chain)
        w3 = w if w.dim() == 3 else w.unsqueeze(0)
        want = w3.double().abs().sum(1).max()
        norm, out = ops_gemm.weight_bound(w, bound, 1.001, 1.0)
        assert abs(float(norm) - float(want)) <= 1e-5 * float(want)
        assert float(out) == pytest.approx(max(1.0, 2.5 * float(norm) * 1.001), rel=1e-6) and out.shape == (1,)
        assert ops_gemm.weight_bound(w)[1] is None
    tiny = torch.full((4, 4), 1e-3, device="cuda")
    assert float(ops_gemm.weight_bound(tiny, bound, 1.001, 1.0)[1]) == 1.0   # the floor
