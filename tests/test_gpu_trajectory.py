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


@pytest.mark.parametrize("gnn,att", [("AT", 3), ("SAGE", 1), ("GCN", 2), ("SAGE", 3)])
def test_parameters_after_three_epochs_match_the_reference(golden_dir, dev, gnn, att):
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
    for ep in range(3):
        lg = utils.resolve_logs(ct.train_step(data, labels.to(dev), ep))          # main.py:313-352: fine-tuning step, then SSL
        cls_logs.append([float(lg[k]) for k in ("loss_train", "acc_train", "loss_reg", "loss_val", "acc_val")])
        logs.append(trs[0].train_step(data)["loss_heads_sup"])
        logs.append(trs[1].train_step(data)["loss_head_disen"])
        logs.append(trs[2].train_step(data)["loss_head_diversity"])
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
