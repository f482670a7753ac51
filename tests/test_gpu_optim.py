"""The multi-tensor Adam launch (csrc/optim.hip, optim.py) against torch.optim.Adam - the optimiser the reference
builds per sub-module (trainer.py:58-60): identical hyper-parameters and independent state per optimiser, parameters
without a gradient skipped, all optimisers of a trainer stepped by one launch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mods(dev, seed):
    torch.manual_seed(seed)
    shapes = [(64, 64), (7,), (1, 1), (130, 33), (4097,), (256, 9), (3, 5, 7), (2048,), (2049,)]
    return [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]


def test_multi_adam_matches_torch_adam():
    from edgedisentangle_ssl_amd import optim
    dev = torch.device("cuda:0")
    ours, ref = _mods(dev, 3), _mods(dev, 3)
    # three "modules" with their own optimiser each, stepped together (what Trainer._finish_step does)
    cuts = [(0, 3), (3, 6), (6, 9)]
    o_opts = [optim.ModuleAdam(ours[a:b], lr=0.01, weight_decay=5e-4) for a, b in cuts]
    r_opts = [torch.optim.Adam(ref[a:b], lr=0.01, weight_decay=5e-4) for a, b in cuts]
    g = torch.Generator(device=dev).manual_seed(9)
    for it in range(25):
        for k, (p, q) in enumerate(zip(ours, ref)):
            if (it + k) % 7 == 3:                       # sometimes a parameter gets no gradient: skipped, step not advanced
                p.grad = q.grad = None
                continue
            gr = torch.randn(p.shape, device=dev, generator=g) * (10.0 ** ((k % 5) - 2))
            p.grad, q.grad = gr.clone(), gr.clone()
        v0 = [p._version for p in ours]
        optim.step_all(o_opts)
        for o in r_opts:
            o.step()
        assert all(p._version > v for p, v in zip(ours, v0) if p.grad is not None)
    for k, (p, q) in enumerate(zip(ours, ref)):
        err = (p - q).abs().max().item()
        assert err <= 2e-6 * max(1.0, q.abs().max().item()), (k, err)
        st = r_opts[[i for i, (a, b) in enumerate(cuts) if a <= k < b][0]].state[q]
        mine = o_opts[[i for i, (a, b) in enumerate(cuts) if a <= k < b][0]].state[id(p)]
        assert int(st["step"]) == mine[0]
        for got, want in ((mine[1], st["exp_avg"]), (mine[2], st["exp_avg_sq"])):     # a few ulp of the tensor's scale
            assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item(), k


def test_multi_adam_many_tensors_and_state_dict():
    """More tensors than one launch's table holds (64), odd sizes, unaligned views are rejected loudly."""
    from edgedisentangle_ssl_amd import optim
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    ours = [torch.nn.Parameter(torch.randn(17 + i, device=dev)) for i in range(150)]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    o, r = optim.ModuleAdam(ours, lr=0.003, weight_decay=0.0), torch.optim.Adam(ref, lr=0.003)
    for it in range(3):
        for p, q in zip(ours, ref):
            p.grad = torch.full_like(p, 0.1 * (it + 1))
            q.grad = p.grad.clone()
        o.step()
        r.step()
    assert max((p - q).abs().max().item() for p, q in zip(ours, ref)) < 1e-6
    sd = o.state_dict()
    o2 = optim.ModuleAdam(ours, lr=1.0)
    o2.load_state_dict(sd)
    assert o2.lr == 0.003 and o2.state[id(ours[5])][0] == 3
    cpu = torch.nn.Parameter(torch.zeros(4))
    cpu.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        optim.ModuleAdam([cpu]).step()


def test_device_step_adam_equals_the_host_step_form():
    """optim.DeviceStepAdam (disgat_adam_multi_dev: step counts in device memory, bias corrections formed in the kernel -
    what a train_step captured in a HIP graph replays) against optim.step_all on the same gradients: identical
    parameters and moments, host step counts kept in step, a re-sync after host-side steps in between."""
    from edgedisentangle_ssl_amd import optim
    dev = torch.device("cuda:0")
    a, b = _mods(dev, 5), _mods(dev, 5)
    cuts = [(0, 4), (4, 9)]
    oa = [optim.ModuleAdam(a[i:j], lr=0.01, weight_decay=5e-4) for i, j in cuts]
    ob = [optim.ModuleAdam(b[i:j], lr=0.01, weight_decay=5e-4) for i, j in cuts]
    dsa = optim.DeviceStepAdam(ob)
    g = torch.Generator(device=dev).manual_seed(2)

    def grads():
        for p, q in zip(a, b):
            gr = torch.randn(p.shape, device=dev, generator=g)
            p.grad, q.grad = gr.clone(), gr.clone()

    for it in range(12):
        grads()
        optim.step_all(oa)
        if it == 6:                       # a host-side step in between (an eager train_step): the device counts re-sync
            optim.step_all(ob)
        else:
            dsa.sync()
            dsa.step()
            dsa.advance_host()
    for k, (p, q) in enumerate(zip(a, b)):
        assert torch.equal(p, q), (k, float((p - q).abs().max()))
    for o1, o2 in zip(oa, ob):
        for p, q in zip(o1.params, o2.params):
            s1, s2 = o1.state[id(p)], o2.state[id(q)]
            assert s1[0] == s2[0] == 12 and torch.equal(s1[1], s2[1]) and torch.equal(s1[2], s2[2])
    assert dsa.steps.tolist() == [12] * len(b)
    # the set of parameters with a gradient must not change under a captured step
    b[0].grad = None
    with pytest.raises(RuntimeError, match="changed"):
        dsa.step()


def test_skinny_linear_rejects_shapes_it_does_not_tile():
    from edgedisentangle_ssl_amd import _lib
    x = torch.zeros(8, 256, device="cuda")
    w = torch.zeros(8, 256, device="cuda")
    y = torch.zeros(8, 8, device="cuda")
    for k, n in ((128, 8), (256, 17), (256, 0)):
        with pytest.raises(RuntimeError):
            _lib.call("disgat_linear_skinny", x.data_ptr(), 256, 8, k, w.data_ptr(), 256, 0, n, y.data_ptr(), 8, 0)
    with pytest.raises(RuntimeError):
        _lib.call("disgat_linear_skinny_wgrad", x.data_ptr(), 256, 8, 256, y.data_ptr(), 8, 8, w.data_ptr(), 6, 0)   # n_waves % 4
