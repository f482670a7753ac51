"""GPU parity of whole optimiser steps: three epochs of the node classifier's step (trainer.py:178-223, with and without the
fusers' L1 term) followed by SupEdge -> DisEdge -> DifHead train steps (main.py:313-352's order) on the tiny graph with
the reference's own pair lists injected, against the parameters the unmodified reference ends with
(tests/golden/tiny_traj_*.npz, written by oracle/gen_golden.py --only traj).  Pins backward + the fused Adam
(trainer.py:58-60: one optimiser per sub-module, lr / weight decay from args, the encoder stepped by every trainer with
that trainer's own moment estimates) and the loss weights (pretrainer.py:750-756) end to end."""
import os

import numpy as np
import pytest
import torch

import inputs_common as ic
from test_gpu_parity import build, close, dev, tiny_inputs  # noqa: F401

pytestmark = pytest.mark.gpu
WEIGHTS = [1.0, 0.5, 2.0]
# Parameters after 12 Adam steps of lr 0.01.  Adam divides by sqrt(v): where a gradient is tiny (the att-1 score weights
# behind the softmax of a sigmoid) its fp32 rounding noise - the reference's as much as ours - moves the parameter by up to
# lr * (relative gradient error) per step, ~1.5e-4 here; everything else agrees to ~1e-6.  A wrong step order, shared
# moments, a missing optimiser or a wrong lr shows up at 1e-3..1e-2.
PTOL = 4e-4


@pytest.mark.parametrize("gnn,att,captured", [("AT", 3, False), ("SAGE", 1, False), ("GCN", 2, False), ("SAGE", 3, False),
                                              ("AT", 3, True), ("SAGE", 1, True), ("GCN", 2, True)])
def test_parameters_after_three_epochs_match_the_reference(golden_dir, dev, gnn, att, captured):
    """captured=True: the same twelve steps replayed from HIP graphs (Trainer.train_step_captured: rolled-back warm-up,
    capture, replays from step 1 on), the golden lists behind the static samplers' interface."""
    import random
    from edgedisentangle_ssl_amd import pretrainer, utils
    from edgedisentangle_ssl_amd.trainer import ClsTrainer
    g = np.load(os.path.join(golden_dir, f"tiny_traj_{gnn}_att{att}.npz"))
    x, adj, n, _ = tiny_inputs(dev)
    idx, _, _ = ic.tiny_graph()
    ci = ic.coalesced_index_set(idx, n)
    labels = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).integers(0, 3, n))
    pos, homo, het = ic.edge_sets(ci, labels, n)
    sup = [t.to(dev) for t in ic.sample_pairs(31, n, pos, "sup")]
    ho = [t.to(dev) for t in ic.sample_pairs(32, n, homo, "homo")]
    he = [t.to(dev) for t in ic.sample_pairs(33, n, het, "het")]
    seed = 100 + att
    a, enc, _ = build(gnn, att, 4, 16, 16, seed, dev)
    a.lr, a.weight_decay, a.dis_type = 0.01, 5e-4, 1
    a.reg, a.reg_weight, a.node_sup_ratio, a.fuse = att != 1, 0.01, 0.25, "last"
    random.seed(5)                      # utils.py:118-151: the class-wise split shuffles with python's `random`
    ct = ClsTrainer(a, enc, labels.to(dev), 1.5)
    for nm in ("idx_train", "idx_val", "idx_test"):
        np.testing.assert_array_equal(getattr(ct, nm).cpu().numpy(), g[nm])
    ic.load_params(ct.fuse1, seed + 41)
    ic.load_params(ct.fuse2, seed + 42)
    ic.load_params(ct.classifier, seed + 43)
    ct.full_metrics = True
    trs = []
    for k, cls in enumerate((pretrainer.SupEdgeTrainer, pretrainer.GeneratedEdgeTrainer, pretrainer.DifHeadTrainer)):
        tr = cls(a, enc, WEIGHTS[k])
        ic.load_params(tr.fuse1, seed + 1 + 10 * k)
        ic.load_params(tr.fuse2, seed + 2 + 10 * k)
        trs.append(tr)
    ic.load_params(trs[2].classifier1, seed + 4)
    ic.load_params(trs[2].classifier2, seed + 5)
    for tr in trs:
        for m in tr.models:
            m.to(dev)
    # the step's sampler replaced by the fixed lists the golden run used
    trs[0].sample_train = lambda gt: (sup[1], [sup[0]])
    trs[1].sample_train = lambda: ([ho[1], he[1]], [ho[0], he[0]])
    data = (x, adj)
    logs, cls_logs = [], []
    if captured:
        from edgedisentangle_ssl_amd import sampling
        from edgedisentangle_ssl_amd.graph import graph_of
        data = (x, graph_of(adj))
        fixed = [sampling.FixedList(p[0], p[1], n) for p in (sup, ho, he)]
        trs[0]._static_sampler = lambda gt: fixed[0]
        trs[1].graph, trs[1]._samplers = data[1], fixed[1:]
        lab_dev = labels.to(dev)
    for ep in range(3):
        if captured:        # the logs are the graph's static outputs: read (or clone) before the next replay
            lg = utils.resolve_logs(ct.train_step_captured(data, lab_dev))
            cls_logs.append([float(lg[k]) for k in ("loss_train", "acc_train", "loss_reg", "loss_val", "acc_val")])
            logs.append(trs[0].train_step_captured(data)["loss_heads_sup"].clone())
            logs.append(trs[1].train_step_captured(data)["loss_head_disen"].clone())
            logs.append(trs[2].train_step_captured(data)["loss_head_diversity"].clone())
            continue
        lg = utils.resolve_logs(ct.train_step(data, labels.to(dev), ep))          # main.py:313-352: fine-tuning step, then SSL
        cls_logs.append([float(lg[k]) for k in ("loss_train", "acc_train", "loss_reg", "loss_val", "acc_val")])
        logs.append(trs[0].train_step(data)["loss_heads_sup"])
        logs.append(trs[1].train_step(data)["loss_head_disen"])
        logs.append(trs[2].train_step(data)["loss_head_diversity"])
    if captured:
        assert all(tr.static_step().replays == 3 and tr.static_step().graph is not None for tr in [ct] + trs)
    np.testing.assert_allclose(np.asarray(cls_logs), g["cls_logs"], rtol=2e-4, atol=2e-5)
    lt = ct.test(data, labels.to(dev))
    # loss / accuracy / sklearn ROC-AUC and macro-F1 of the test split (trainer.py:296-318)
    np.testing.assert_allclose([lt["loss_test"], lt["acc_test"], lt["roc_test"], lt["macroF_test"]], g["cls_test"],
                               rtol=2e-4, atol=2e-5)
    got = torch.stack([torch.as_tensor(v).float().reshape(()) for v in logs]).cpu().numpy()
    np.testing.assert_allclose(got, g["losses"], rtol=2e-4, atol=2e-5)      # the reference rounds its logs to 5 decimals
    checked = 0
    for k, p in enc.state_dict().items():
        close(p, g["enc." + k], tol=PTOL, what="enc." + k)
        checked += 1
    for t, tr in enumerate(trs):
        for nm in ("fuse1", "fuse2"):
            for k, p in getattr(tr, nm).state_dict().items():
                close(p, g[f"t{t}.{nm}.{k}"], tol=PTOL, what=f"t{t}.{nm}.{k}")
                checked += 1
    for nm in ("classifier1", "classifier2"):
        for k, p in getattr(trs[2], nm).state_dict().items():
            close(p, g[f"t2.{nm}.{k}"], tol=PTOL, what=f"t2.{nm}.{k}")
            checked += 1
    for nm in ("fuse1", "fuse2", "classifier"):
        for k, p in getattr(ct, nm).state_dict().items():
            close(p, g[f"cls.{nm}.{k}"], tol=PTOL, what=f"cls.{nm}.{k}")
            checked += 1
    assert checked == len(g.files) - 6          # losses, cls_logs, cls_test and the three index sets


@pytest.mark.parametrize("name,gnn,att", [("cora", "AT", 3), ("cora", "SAGE", 1), ("chameleon", "AT", 3), ("chameleon", "SAGE", 1),
                                          ("chameleon", "SAGE", 3)])
def test_real_graph_parameters_after_two_epochs_match_the_reference(golden_dir, dev, name, gnn, att):
    """The same pin at the real configurations (BASELINE configs[0] / [1]): Cora (bundled adjacency, seeded surrogate
    features) and chameleon (real features), H = 8, nhid = 64, two epochs of CLS -> SupEdge -> DisEdge -> DifHead with the
    reference's pair lists injected (tests/golden/<graph>_traj_*.npz, oracle/gen_golden.py --only traj_real: the first
    256 rows of every parameter plus its sum and abs-sum).

    Tolerances are conditioning-aware.  Adam divides by sqrt(v): an element whose gradient is a cancelling sum at fp32
    noise level takes a +-lr step of either sign, in the reference itself.  The generator therefore runs the reference a
    second time on inputs perturbed by ~2 ulp and records, per parameter, how far it moves (`#sens`: up to 2e-2 = two
    full steps) and how many elements move by more than PTOL (`#nflip`: 0.2-2 % of W_em / fuse / classifier weights),
    and the same for every logged loss.  Held here: the bulk of every parameter tightly (median |d| <= 1e-4; measured
    1e-8..3e-5), the number of outlying elements to the reference's own count (x4, or 2.5 % of the tensor), no element
    further than the 8 steps x lr it can move at all, losses to 2e-4 + 4 x their recorded sensitivity.  A wrong step
    order, shared moments, a missing optimiser or a wrong lr moves EVERY element by 1e-3..1e-2.
    Round 5: SAGE on chameleon too (N(0,1) weights on features of magnitude 9e2, losses ~5e2: the reference's own 2-ulp
    perturbation moves single parameters by up to 4.5e-2 and the logged losses by up to 9e-2 - its recorded sensitivity is
    the yardstick, as for the other graphs)."""
    import random
    from edgedisentangle_ssl_amd import pretrainer, utils
    from edgedisentangle_ssl_amd.trainer import ClsTrainer
    from test_gpu_parity import real_inputs
    g = np.load(os.path.join(golden_dir, f"{name}_traj_{gnn}_att{att}.npz"))
    x, adj, n, ei, sup, ho, he = real_inputs(golden_dir, name, dev)
    labels = torch.from_numpy(np.load(os.path.join(golden_dir, f"data_{name}.npz"))["labels"].astype(np.int64))
    sup, ho, he = ([t.to(dev) for t in p] for p in (sup, ho, he))
    seed = 300 + att
    a, enc, _ = build(gnn, att, 8, 64, 64, seed, dev)
    a.lr, a.weight_decay, a.dis_type = 0.01, 5e-4, 1
    a.reg, a.reg_weight, a.node_sup_ratio, a.fuse = att != 1, 0.01, 0.25, "last"
    random.seed(5)
    ct = ClsTrainer(a, enc, labels.to(dev), 1.5)
    for nm in ("idx_train", "idx_val", "idx_test"):
        np.testing.assert_array_equal(getattr(ct, nm).cpu().numpy(), g[nm])
    ic.load_params(ct.fuse1, seed + 41)
    ic.load_params(ct.fuse2, seed + 42)
    ic.load_params(ct.classifier, seed + 43)
    ct.full_metrics = True
    trs = []
    for k, cls in enumerate((pretrainer.SupEdgeTrainer, pretrainer.GeneratedEdgeTrainer, pretrainer.DifHeadTrainer)):
        tr = cls(a, enc, WEIGHTS[k])
        ic.load_params(tr.fuse1, seed + 1 + 10 * k)
        ic.load_params(tr.fuse2, seed + 2 + 10 * k)
        trs.append(tr)
    ic.load_params(trs[2].classifier1, seed + 4)
    ic.load_params(trs[2].classifier2, seed + 5)
    for tr in trs:
        for m in tr.models:
            m.to(dev)
    trs[0].sample_train = lambda gt: (sup[1], [sup[0]])
    trs[1].sample_train = lambda: ([ho[1], he[1]], [ho[0], he[0]])
    data = (x, adj)
    logs, cls_logs = [], []
    for ep in range(2):
        lg = utils.resolve_logs(ct.train_step(data, labels.to(dev), ep))
        cls_logs.append([float(lg[k]) for k in ("loss_train", "acc_train", "loss_reg", "loss_val", "acc_val")])
        logs.append(trs[0].train_step(data)["loss_heads_sup"])
        logs.append(trs[1].train_step(data)["loss_head_disen"])
        logs.append(trs[2].train_step(data)["loss_head_diversity"])
    np.testing.assert_allclose(np.asarray(cls_logs)[0], g["cls_logs"][0], rtol=2e-5, atol=2e-6)     # same parameters on both sides
    cl = np.asarray(cls_logs)
    assert np.all(np.abs(cl - g["cls_logs"]) <= 1e-3 * np.abs(g["cls_logs"]) + 5e-5 + 4 * g["cls_logs#sens"]), (cl, g["cls_logs"])
    got = torch.stack([torch.as_tensor(v).float().reshape(()) for v in logs]).cpu().numpy().astype(np.float64)
    # (the reference rounds its logs to 5 decimals)
    assert np.all(np.abs(got[:3] - g["losses"][:3]) <= 2e-4 * np.abs(g["losses"][:3]) + 2e-5 + 4 * g["losses#sens"][:3]), (got, g["losses"])
    assert np.all(np.abs(got - g["losses"]) <= 1e-3 * np.abs(g["losses"]) + 2e-5 + 4 * g["losses#sens"]), (got, g["losses"])
    checked = 0

    def same(p, key):
        nonlocal checked
        pc = p.detach().cpu().double()
        head = pc[:256] if pc.dim() else pc
        d = (head - torch.from_numpy(np.asarray(g[key], dtype=np.float64))).abs()
        assert torch.isfinite(pc).all(), key
        # (chameleon: the score parameters sit behind a saturated sigmoid - raw scores ~1e3 - and their whole gradient is noise)
        # (SAGE on chameleon: the reference's own median movement under the 2-ulp perturbation is recorded, `#med`.  Another
        # fp32 evaluation order differs from the reference's by ~1e-6 relative, several times the probe's perturbation -
        # measured here: medians up to 9.5x and outlier counts up to 4.1x the reference's own; held to 16x / 8x.  What pins
        # this combination is above: the first CLS log to 2e-5 and every loss to its recorded sensitivity.)
        loose = key + "#med" in g.files
        med_ref = float(g[key + "#med"]) if loose else 0.0
        assert float(d.median()) <= max(4e-4 if name == "chameleon" else 1e-4, 16 * med_ref), (key, "median", float(d.median()), med_ref)
        assert float(d.max()) <= 8 * 0.01 * 1.05, (key, "max", float(d.max()))
        n_out = int((d > PTOL).sum())
        assert n_out <= max((8 if loose else 4) * int(g[key + "#nflip"]), d.numel() // 40, 8), (key, "outliers", n_out, int(g[key + "#nflip"]), d.numel())
        checked += 1

    for k, p in enc.state_dict().items():
        same(p, "enc." + k)
    for t, tr in enumerate(trs):
        for nm in ("fuse1", "fuse2"):
            for k, p in getattr(tr, nm).state_dict().items():
                same(p, f"t{t}.{nm}.{k}")
    for nm in ("classifier1", "classifier2"):
        for k, p in getattr(trs[2], nm).state_dict().items():
            same(p, f"t2.{nm}.{k}")
    for nm in ("fuse1", "fuse2", "classifier"):
        for k, p in getattr(ct, nm).state_dict().items():
            same(p, f"cls.{nm}.{k}")
    per = 6 if any(k.endswith("#med") for k in g.files) else 5
    assert per * checked == len(g.files) - 7        # losses, cls_logs, their sensitivities and the three index sets
