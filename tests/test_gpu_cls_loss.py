"""disgat_cls_loss / _bwd (csrc/cls_loss.hip) against the ATen formulation the reference runs: F.log_softmax -> F.nll_loss on
the training split + utils.accuracy, the same on the validation split (trainer.py:186-199), and DifHead's NLL against the
head index (pretrainer.py:819-832).  fp32 tolerance 1e-6 relative on the values (sums are taken in double here), 1e-6
absolute on the gradients; counts exact."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(n, c, seed, frac=(0.3, 0.2), pad=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    store = torch.randn(n, c + pad, device="cuda", generator=g) * 3.0
    logits = store[:, :c]                                            # row stride c + pad
    labels = torch.randint(0, c, (n,), device="cuda", generator=g)
    perm = torch.randperm(n, device="cuda", generator=g)
    n_tr, n_va = int(n * frac[0]), int(n * frac[1])
    idx_tr, idx_va = perm[:n_tr], perm[n_tr:n_tr + n_va]
    code = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    code[idx_tr] = labels[idx_tr].int()
    code[idx_va] = labels[idx_va].int() + 65536
    return logits, labels, idx_tr, idx_va, code


@pytest.mark.parametrize("n,c,pad", [(2277, 5, 0), (2708, 7, 1), (1, 3, 0), (8192, 70, 0), (8193, 4, 0), (300_000, 10, 2),
                                     (5000, 16, 0), (3000, 8, 4), (20_000, 8, 0), (777, 4, 1)])
def test_matches_log_softmax_nll_accuracy(n, c, pad):
    from edgedisentangle_ssl_amd import ops
    logits, labels, idx_tr, idx_va, code = _case(n, c, 100 + n % 97, pad=pad)
    n_tr, n_va = max(1, idx_tr.numel()), max(1, idx_va.numel())
    x = logits.detach().clone().requires_grad_(True) if pad == 0 else None
    src = x if x is not None else logits.detach().requires_grad_(False)
    loss, logp, res = ops.cls_loss(src, code, 0, n_tr, n_va)
    ref_in = logits.detach().clone().requires_grad_(True)
    ref_lp = F.log_softmax(ref_in, dim=1)
    assert torch.allclose(logp, ref_lp, rtol=0, atol=2e-6)
    for k, idx, nn in ((0, idx_tr, n_tr), (2, idx_va, n_va)):
        if idx.numel() == 0:
            assert float(res[k]) == 0.0 and float(res[k + 1]) == 0.0
            continue
        want = F.nll_loss(ref_lp[idx].double(), labels[idx])
        assert abs(float(res[k]) - float(want.detach())) <= 1e-6 * max(1.0, abs(float(want.detach())))
        correct = int((ref_lp[idx].argmax(1) == labels[idx]).sum())
        assert float(res[k + 1]) == correct / nn
    assert float(loss) == pytest.approx(float(res[0]), rel=1e-6)
    if x is not None and idx_tr.numel():
        (loss * 1.7).backward()
        (F.nll_loss(ref_lp[idx_tr], labels[idx_tr]) * 1.7).backward()
        assert torch.allclose(x.grad, ref_in.grad, rtol=1e-5, atol=1e-7 + 2e-6 / n_tr)
        assert bool((x.grad[code != code.clamp(0, 65535)] == 0).all())        # validation rows and rows in no split: zeros
    # the same call again: bit-identical (fixed-order double sums)
    loss2, _, res2 = ops.cls_loss(src.detach(), code, 0, n_tr, n_va)
    assert torch.equal(res, res2) and torch.equal(loss.detach(), loss2)


@pytest.mark.parametrize("n,nh", [(2277, 8), (3000, 4), (100_000, 8)])
def test_head_index_labels(n, nh):
    """DifHead: rows are (node, head) pairs, the label of row r is r % nh; mean over nodes, sum over heads."""
    from edgedisentangle_ssl_amd import ops
    g = torch.Generator(device="cuda").manual_seed(n)
    t = (torch.randn(n * nh, nh, device="cuda", generator=g) * 2.0).requires_grad_(True)
    loss, _, res = ops.cls_loss(t, None, nh, n)
    ref_in = t.detach().clone().requires_grad_(True)
    want = -F.log_softmax(ref_in, dim=1).view(-1, nh, nh).diagonal(dim1=1, dim2=2).mean(0).sum()
    assert float(loss) == pytest.approx(float(want), rel=2e-6)
    loss.backward()
    want.backward()
    assert torch.allclose(t.grad, ref_in.grad, rtol=1e-5, atol=1e-9 + 2e-6 / n)


def test_ties_take_the_first_maximum_and_bad_arguments_raise():
    from edgedisentangle_ssl_amd import ops
    x = torch.zeros(4, 3, device="cuda")
    code = torch.tensor([0, 1, 2, -1], dtype=torch.int32, device="cuda")
    _, _, res = ops.cls_loss(x, code, 0, 3, 1)
    assert float(res[1]) == pytest.approx(1 / 3)                      # argmax of equal logits is class 0
    assert float(res[0]) == pytest.approx(float(torch.log(torch.tensor(3.0))), rel=1e-6)
    with pytest.raises(RuntimeError):
        ops.cls_loss(x.cpu(), code, 0, 3, 1)
    with pytest.raises(RuntimeError):
        ops.cls_loss(x, code.long(), 0, 3, 1)
    with pytest.raises(RuntimeError):
        ops.cls_loss(x, None, 5, 3, 1)                               # head labels beyond the class count
    with pytest.raises(RuntimeError):
        ops.cls_loss(x, code, 0, 0, 1)                               # a zero divisor
