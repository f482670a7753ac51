"""train_steps replayed from HIP graphs (capture.StaticStep; main.run --capture on).

* The captured run follows the SAME trajectory as the same static step executed eagerly: two identical trainer sets on
  Cora (dropout 0.1 on the input and on the attention, pair lists from the fixed-capacity device sampler), one stepping
  through StaticStep.run_eager, one through train_step_captured (rolled-back warm-up, capture, replays) - every
  parameter must agree bit for bit after four epochs, which pins the roll-back, the device-side Adam step counts, the
  dropout seed counter and the RNG offsets a replay consumes.
* The reference's golden trajectories on the captured path: tests/test_gpu_trajectory.py (captured=True cases).
* main.run --capture on trains (finite, decreasing losses) and agrees statistically with --capture off."""
import math
import os

import pytest
import torch

import inputs_common as ic

pytestmark = pytest.mark.gpu


def _trainers(golden_dir, dev, seed):
    import random
    from edgedisentangle_ssl_amd import pretrainer
    from edgedisentangle_ssl_amd.graph import graph_of
    from edgedisentangle_ssl_amd.trainer import ClsTrainer
    from test_gpu_parity import build, real_inputs
    import numpy as np
    x, adj, n, _ei, _s, _h, _e = real_inputs(golden_dir, "cora", dev)
    labels = torch.from_numpy(np.load(os.path.join(golden_dir, "data_cora.npz"))["labels"].astype(np.int64)).to(dev)
    a, enc, _ = build("AT", 3, 4, 32, x.shape[1], 77, dev)
    a.lr, a.weight_decay, a.dis_type, a.dropout = 0.01, 5e-4, 1, 0.1
    a.reg, a.reg_weight, a.node_sup_ratio, a.fuse = True, 0.01, 0.25, "last"
    for m in enc.modules():
        if hasattr(m, "dropout"):
            m.dropout = 0.1
    random.seed(5)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    g = graph_of(adj)
    ct = ClsTrainer(a, enc, labels, 1.0)
    trs = [pretrainer.SupEdgeTrainer(a, enc, 1.0), pretrainer.GeneratedEdgeTrainer(a, enc, 0.5), pretrainer.DifHeadTrainer(a, enc, 2.0)]
    for k, tr in enumerate([ct] + trs):
        ic.load_params(tr.fuse1, 500 + 2 * k)
        ic.load_params(tr.fuse2, 501 + 2 * k)
        for m in tr.models:
            m.to(dev)
    trs[1].get_label_all(x, g, labels)
    return (x, g), labels, ct, trs


def _params(ct, trs):
    out = {}
    for k, tr in enumerate([ct] + trs):
        for mi, m in enumerate(tr.models):
            for name, p in m.state_dict().items():
                out[f"t{k}.m{mi}.{name}"] = p.detach().clone()
    return out


def test_captured_run_equals_the_eager_static_run(golden_dir):
    dev = torch.device("cuda")
    runs = []
    for captured in (False, True):
        data, labels, ct, trs = _trainers(golden_dir, dev, seed=9)
        losses = []
        for _ep in range(4):
            for _ in range(2):
                lg = (ct.train_step_captured(data, labels) if captured else ct.static_step().run_eager(data, labels))
                losses.append(lg["loss_train"].clone())
            for tr, extra in ((trs[0], (data[1],)), (trs[1], ()), (trs[2], ())):
                lg = tr.train_step_captured(data, *extra) if captured else tr.static_step().run_eager(data, *extra)
                losses.append(next(iter(lg.values())).clone())
        if captured:
            assert all(t.static_step().graph is not None for t in [ct] + trs)
            assert ct.static_step().replays == 8 and trs[0].static_step().replays == 4
        smp = data[1].__dict__["_pair_sampler"]
        assert smp.events() == (0, 0) and int(smp.meta[1]) == 4           # one list per SupEdge step, none for warm-up or capture
        runs.append((torch.stack(losses).cpu(), _params(ct, trs), [o.state_dict() for o in ct.models_opt]))
    assert torch.isfinite(runs[0][0]).all()
    assert torch.equal(runs[0][0], runs[1][0]), (runs[0][0], runs[1][0])
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k
    for oa, ob in zip(runs[0][2], runs[1][2]):       # host step counts and moments follow the replays
        for sa, sb in zip(oa["state"], ob["state"]):
            assert (sa is None) == (sb is None)
            if sa is not None:
                assert sa["step"] == sb["step"] == 8 and torch.equal(sa["exp_avg"], sb["exp_avg"])


def test_a_captured_step_refuses_other_tensors(golden_dir):
    dev = torch.device("cuda")
    data, labels, ct, _trs = _trainers(golden_dir, dev, seed=3)
    ct.train_step_captured(data, labels)
    with pytest.raises(RuntimeError, match="captured"):
        ct.train_step_captured((data[0].clone(), data[1]), labels)


@pytest.mark.parametrize("gnn,att", [("AT", 3), ("SAGE", 1), ("GCN", 2)])
def test_main_flow_trains_captured(golden_dir, gnn, att):
    from edgedisentangle_ssl_amd import main
    argv = ["--model=DISGAT", "--sparse", "--dataset", "chameleon", "--fixture", os.path.join(golden_dir, "data_chameleon.npz"),
            "--gnn_type", gnn, "--att", str(att), "--nhead", "4", "--nhid", "32", "--epochs", "8", "--steps", "2",
            "--downstream", "CLS", "--down_weight", "1.0", "--finetune", "--pretrain", "SupEdge", "DisEdge", "DifHead",
            "--pre_weight", "1", "1", "1", "--pre_edge", "1", "1", "1", "--dropout", "0.1", "--seed", "4", "--quiet"]
    hist = main.run(argv + ["--capture", "on"])
    ref = main.run(argv + ["--capture", "off"])
    assert len(hist) == 8
    for h in hist:
        for k in ("loss_train", "loss_heads_sup", "loss_head_disen", "loss_head_diversity"):
            assert math.isfinite(h[k]), (k, h)
    assert hist[-1]["loss_head_diversity"] < hist[0]["loss_head_diversity"]
    assert hist[-1]["loss_train"] < hist[0]["loss_train"]
    # different dropout masks and pair lists, same distribution: the two runs stay close
    for k in ("loss_train", "loss_head_diversity"):
        assert abs(hist[-1][k] - ref[-1][k]) < 0.25 * abs(ref[-1][k]) + 0.05, (k, hist[-1][k], ref[-1][k])


def test_graphs_die_with_their_trainers_by_reference_count(golden_dir):
    """trainer and StaticStep used to reference each other, so HIP graphs were freed only by the cyclic
    collector, at a time nobody chose - once inside a later capture, where destroying a graph aborts the process.  Now the
    step holds its trainer weakly: dropping the trainers frees steps and graphs on the spot (checked with the collector
    OFF), Trainer.close() does the same explicitly, and a second trainer set captures and replays with the collector
    running at every allocation (gc.set_threshold(1))."""
    import gc
    import weakref
    dev = torch.device("cuda")
    data, labels, ct, trs = _trainers(golden_dir, dev, seed=21)
    ct.train_step_captured(data, labels)
    for tr, extra in ((trs[0], (data[1],)), (trs[1], ()), (trs[2], ())):
        tr.train_step_captured(data, *extra)
    steps = [weakref.ref(t.static_step()) for t in [ct] + trs]
    graphs = [weakref.ref(t.static_step().graph) for t in [ct] + trs]
    owners = [weakref.ref(t) for t in [ct] + trs]
    assert all(g() is not None for g in graphs)
    gc.collect()
    gc.disable()
    try:
        trs[2].close()                                  # explicit: the graph goes, the trainer stays usable
        assert graphs[3]() is None and steps[3]() is None and owners[3]() is not None
        del ct, trs, tr
        assert all(r() is None for r in steps + graphs + owners), [r() for r in steps + graphs + owners]
    finally:
        gc.enable()
    old = gc.get_threshold()
    gc.set_threshold(1)
    try:
        data2, labels2, ct2, trs2 = _trainers(golden_dir, dev, seed=22)
        for _ in range(2):
            ct2.train_step_captured(data2, labels2)
            for tr, extra in ((trs2[0], (data2[1],)), (trs2[1], ()), (trs2[2], ())):
                lg = tr.train_step_captured(data2, *extra)
        torch.cuda.synchronize()
        assert all(torch.isfinite(v).all() for v in lg.values())
    finally:
        gc.set_threshold(*old)
