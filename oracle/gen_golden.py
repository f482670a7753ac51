#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED
reference (read-only at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (the reference never
travels to the GPU box); the product never imports this.  The reference's .py
files are imported in place with an in-memory `ipdb` stub; nothing
is copied out of them - fixtures hold inputs-by-seed and expected OUTPUTS only.

Usage:  python oracle/gen_golden.py [--only tiny|prims|traj|traj_real|labels|real|grads_real|tiny256|rows]
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import scipy.sparse as sp
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"

sys.dont_write_bytecode = True
sys.modules["ipdb"] = types.ModuleType("ipdb")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REPO, "tests"))

import utils as ref_utils          # noqa: E402  (reference)
import layers as ref_layers        # noqa: E402
import models as ref_models        # noqa: E402
import pretrainer as ref_pre       # noqa: E402
import inputs_common as ic         # noqa: E402  (ours)

GNNS = ["AT", "SAGE", "GCN"]
ATTS = [1, 2, 3]


def ref_args(gnn, att, nhead, nhid, size, extra=()):
    a = ref_utils.get_parser().parse_args(
        ["--model", "DISGAT", "--sparse", "--dropout", "0", "--att", str(att), "--gnn_type", gnn,
         "--nhead", str(nhead), "--nhid", str(nhid), "--dataset", "golden", *extra])
    a.cuda = False
    a.size = size
    a.hetero = True
    return a


def sparse_adj(indices, values, n):
    return torch.sparse_coo_tensor(indices, values, (n, n))


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def inject_sampler(trainer, ret):
    trainer.sample_train = lambda *a, **k: ret


def grads_of(prefix, module, out):
    for k, p in module.named_parameters():
        out[f"{prefix}.{k}"] = np32(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), np.float32)


# --------------------------------------------------------------------------- prims
def gen_prims():
    out = {}
    idx, vals, n = ic.tiny_graph()
    ci = ic.coalesced_index_set(idx, n)
    g = np.random.Generator(np.random.PCG64(5))
    v = torch.from_numpy(g.standard_normal((ci.shape[1], 1)).astype(np.float32))
    mat = torch.from_numpy(g.standard_normal((n, 12)).astype(np.float32))
    out["softmax_in"] = np32(v)
    out["mat"] = np32(mat)
    sm = ref_utils.sp_softmax(ci, v, n)                       # utils.py:192
    out["softmax_out"] = np32(sm)
    out["matmul_out"] = np32(ref_utils.sp_matmul(ci, sm, mat))  # utils.py:203
    # adj_mse_loss 1-D quirk and 2-D regular case (utils.py:287)
    rec = torch.from_numpy(g.random(500).astype(np.float32))
    tgt = torch.from_numpy((g.random(500) < 0.3).astype(np.float32))
    out["mse_rec"], out["mse_tgt"] = np32(rec), np32(tgt)
    out["mse_1d"] = np32(ref_utils.adj_mse_loss(rec, tgt))
    out["mse_2d"] = np32(ref_utils.adj_mse_loss(rec[:400].reshape(20, 20), tgt[:400].reshape(20, 20)))
    # FuseLayer variants (layers.py:876-921)
    feats = [torch.from_numpy(g.standard_normal((n, 8)).astype(np.float32)) for _ in range(4)]
    res = torch.from_numpy(g.standard_normal((n, 10)).astype(np.float32))
    for k, f in enumerate(feats):
        out[f"fuse_in{k}"] = np32(f)
    out["fuse_res"] = np32(res)
    for rt in (0, 1, 2):
        for norelu in (0, 1):
            for use_res in (0, 1):
                extra = ["--residue_type", str(rt)] + (["--fuse_no_relu"] if norelu else []) + (["--residue"] if use_res else [])
                a = ref_args("AT", 1, 4, 8, 10, extra)
                fl = ref_layers.FuseLayer(a, 4, nfeat=8, residue=10 if use_res else 0)
                ic.load_params(fl, 70 + rt)
                out[f"fuse_rt{rt}_nr{norelu}_res{use_res}"] = np32(fl(feats, res))
    # MLP (models.py:523)
    mlp = ref_models.MLP(in_feat=8, hidden_size=6, out_size=4, layers=2)
    ic.load_params(mlp, 80)
    out["mlp_raw"] = np32(mlp(feats[0]))
    out["mlp_cls"] = np32(mlp(feats[0], cls=True))
    # utils.group_correlation (utils.py:326), used by Trainer.analyze_disentangle
    emb = torch.from_numpy(g.standard_normal((7, 33)).astype(np.float32))
    out["corr_in"] = np32(emb)
    out["corr_out"] = np32(ref_utils.group_correlation(emb))
    np.savez_compressed(os.path.join(GOLD, "prims.npz"), **out)
    print("prims.npz", len(out))
    # initial parameters of the reference's DISGAT under torch.manual_seed(4) (RNG draw order contract)
    init = {}
    for gnn in GNNS:
        for att in (1, 3):
            a = ref_args(gnn, att, 3, 12, 20)
            torch.manual_seed(4)
            m = ref_models.DISGAT(a, nfeat=20, nhid=12, nclass=12, nheads=3, dropout=0.1)
            for k, v in m.state_dict().items():
                init[f"{gnn}_{att}.{k}"] = np32(v)
    np.savez_compressed(os.path.join(GOLD, "init_seed4.npz"), **init)
    print("init_seed4.npz", len(init))


# --------------------------------------------------------------------------- one (gnn, att) case
def run_case(x, adj, n, labels, gnn, att, nhead, nhid, seed, aux_lists, sup, dis, store_grads, out, pre=""):
    """Runs every boundary entry point + the three losses for one combo.
    sup = (labels[M], [indices]); dis = ([lab_homo, lab_het], [idx_homo, idx_het])."""
    size = x.shape[1]
    a = ref_args(gnn, att, nhead, nhid, size)

    def fresh_encoder():
        m = ref_models.DISGAT(a, nfeat=size, nhid=nhid, nclass=nhid, nheads=nhead, dropout=0.0)
        return ic.load_params(m, seed)

    fus = [ic.load_params(ref_layers.FuseLayer(a, nhead, nfeat=nhid), seed + 1),
           ic.load_params(ref_layers.FuseLayer(a, nhead, nfeat=nhid), seed + 2)]
    enc = fresh_encoder().eval()
    with torch.no_grad():
        out[pre + "forward"] = np32(enc(x, adj, fus))                         # models.py:181
        em = enc.get_em(x, adj, fus)                                          # models.py:217
        out[pre + "get_em_0"], out[pre + "get_em_1"] = np32(em[0]), np32(em[1])
        adjs = enc.get_adjs(x, adj, fus)                                      # models.py:254
        for l in range(2):
            out[pre + f"adjs_{l}"] = np.stack([np32(t)[:, 0] for t in adjs[l]])          # [H,E]
        aux = enc.predict_adjs_sparse(x, adj, fus, aux_lists)                 # models.py:290
        for l in range(2):
            for j in range(len(aux_lists)):
                out[pre + f"aux_{l}_{j}"] = np.stack([np32(h[j])[:, 0] for h in aux[l]])  # [H,M_j]
        ee = enc.get_edge_em(x, adj, fus)                                     # models.py:333
        for l in range(2):
            out[pre + f"edge_em_{l}"] = np.stack([np32(t) for t in ee[l]])    # [H,N,F+nhid]

    # single layer with aux (layers.py:493)
    lay = ref_layers.DisGALayer(size, nhid, dropout=0.0, alpha=0.1, concat=True, att_type=att, gnn_type=gnn)
    ic.load_params(lay, seed + 3).eval()
    with torch.no_grad():
        h, e, au = lay(x, adj, aux_lists)
    out[pre + "layer_h"], out[pre + "layer_e"] = np32(h), np32(e)[:, 0]
    for j, t in enumerate(au):
        out[pre + f"layer_aux_{j}"] = np32(t)[:, 0]

    # --- losses through the reference trainers' own train_step (pretrainer.py:709, 578, 810)
    def prep(tr, s):
        ic.load_params(tr.fuse1, s + 1)
        ic.load_params(tr.fuse2, s + 2)

    enc = fresh_encoder()
    tr = ref_pre.SupEdgeTrainer(a, enc, 1.0)
    prep(tr, seed)
    inject_sampler(tr, sup)
    log = tr.train_step((x, adj), None)
    out[pre + "loss_sup"] = np.float32(log["loss_heads_sup"])
    if store_grads:
        grads_of(pre + "gsup.enc", enc, out)
        grads_of(pre + "gsup.fuse1", tr.fuse1, out)

    enc = fresh_encoder()
    tr = ref_pre.GeneratedEdgeTrainer(a, enc, 1.0)
    prep(tr, seed)
    tr.dis_adjs = [None, None]
    inject_sampler(tr, dis)
    log = tr.train_step((x, adj))
    out[pre + "loss_dis"] = np.float32(log["loss_head_disen"])
    if store_grads:
        grads_of(pre + "gdis.enc", enc, out)

    enc = fresh_encoder()
    tr = ref_pre.DifHeadTrainer(a, enc, 1.0)
    prep(tr, seed)
    ic.load_params(tr.classifier1, seed + 4)
    ic.load_params(tr.classifier2, seed + 5)
    log = tr.train_step((x, adj))
    out[pre + "loss_dif"] = np.float32(log["loss_head_diversity"])
    if store_grads:
        grads_of(pre + "gdif.enc", enc, out)
        grads_of(pre + "gdif.cls1", tr.classifier1, out)
        grads_of(pre + "gdif.fuse1", tr.fuse1, out)
        grads_of(pre + "gdif.fuse2", tr.fuse2, out)


# --------------------------------------------------------------------------- tiny
def gen_tiny():
    idx, vals, n = ic.tiny_graph()
    adj = sparse_adj(idx, vals, n)
    x = ic.features(21, n, 16)
    labels = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).integers(0, 3, n))
    ci = ic.coalesced_index_set(idx, n)
    pos, homo, het = ic.edge_sets(ci, labels, n)
    sup_idx, sup_lab = ic.sample_pairs(31, n, pos, "sup")
    ho_idx, ho_lab = ic.sample_pairs(32, n, homo, "homo")
    he_idx, he_lab = ic.sample_pairs(33, n, het, "het")
    aux_lists = [ic.aux_pairs(41, n, 300, "a0"), ic.aux_pairs(42, n, 150, "a1")]
    for gnn in GNNS:
        for att in ATTS:
            out = {}
            run_case(x, adj, n, labels, gnn, att, 4, 16, 100 + att, aux_lists,
                     (sup_lab, [sup_idx]), ([ho_lab, he_lab], [ho_idx, he_idx]), True, out)
            np.savez_compressed(os.path.join(GOLD, f"tiny_{gnn}_att{att}.npz"), **out)
            print("tiny", gnn, att, "sup %.6f dis %.6f dif %.6f" % (out["loss_sup"], out["loss_dis"], out["loss_dif"]))

    # The reference's OWN samplers on the tiny graph (recorded: they use torch/np RNG)
    a = ref_args("AT", 3, 4, 16, 16)
    enc = ref_models.DISGAT(a, nfeat=16, nhid=16, nclass=16, nheads=4, dropout=0.0)
    out = {}
    tr = ref_pre.SupEdgeTrainer(a, enc, 1.0)
    gt = tr.get_label_all(x, adj)                      # pretrainer.py:667
    torch.manual_seed(6); np.random.seed(6)
    lab, ind = tr.sample_train(gt)                     # pretrainer.py:683
    out["sup_idx"], out["sup_lab"] = ind[0].numpy().astype(np.int16), np32(lab)
    tr = ref_pre.GeneratedEdgeTrainer(a, enc, 1.0)
    tr.get_label_all(x, adj, labels, load=False)       # pretrainer.py:386
    torch.manual_seed(6); np.random.seed(6)
    labs, inds = tr.sample_train()                     # pretrainer.py:524
    for j in range(2):
        out[f"dis_idx{j}"], out[f"dis_lab{j}"] = inds[j].numpy().astype(np.int16), np32(labs[j])
    out["labels"] = labels.numpy().astype(np.int16)
    np.savez_compressed(os.path.join(GOLD, "tiny_ref_sampler.npz"), **out)
    print("tiny_ref_sampler", {k: v.shape for k, v in out.items()})


def gen_trajectory():
    """Several optimiser steps of the reference's own trainers (pretrainer.py:709-763, 578-641, 810-847 with
    trainer.py:58-60's per-module Adam) on the tiny graph with injected pair lists, dropout 0: the parameters afterwards
    pin backward + Adam (lr, weight decay, one optimiser per sub-module, the encoder stepped by every trainer) end to end."""
    idx, vals, n = ic.tiny_graph()
    adj = sparse_adj(idx, vals, n)
    x = ic.features(21, n, 16)
    labels = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).integers(0, 3, n))
    ci = ic.coalesced_index_set(idx, n)
    pos, homo, het = ic.edge_sets(ci, labels, n)
    sup_idx, sup_lab = ic.sample_pairs(31, n, pos, "sup")
    ho_idx, ho_lab = ic.sample_pairs(32, n, homo, "homo")
    he_idx, he_lab = ic.sample_pairs(33, n, het, "het")
    for gnn, att in (("AT", 3), ("SAGE", 1), ("GCN", 2), ("SAGE", 3)):
        seed = 100 + att
        a = ref_args(gnn, att, 4, 16, 16, extra=("--reg",) if att != 1 else ())
        a.lr, a.weight_decay = 0.01, 5e-4
        enc = ic.load_params(ref_models.DISGAT(a, nfeat=16, nhid=16, nclass=16, nheads=4, dropout=0.0), seed)
        import random
        import trainer as ref_trainer
        random.seed(5)                                            # utils.py:134,151: the split shuffles with `random`
        ct = ref_trainer.ClsTrainer(a, enc, labels, 1.5)
        ic.load_params(ct.fuse1, seed + 41)
        ic.load_params(ct.fuse2, seed + 42)
        ic.load_params(ct.classifier, seed + 43)
        trs = []
        for k, cls in enumerate((ref_pre.SupEdgeTrainer, ref_pre.GeneratedEdgeTrainer, ref_pre.DifHeadTrainer)):
            tr = cls(a, enc, [1.0, 0.5, 2.0][k])
            ic.load_params(tr.fuse1, seed + 1 + 10 * k)
            ic.load_params(tr.fuse2, seed + 2 + 10 * k)
            trs.append(tr)
        ic.load_params(trs[2].classifier1, seed + 4)
        ic.load_params(trs[2].classifier2, seed + 5)
        inject_sampler(trs[0], (sup_lab, [sup_idx]))
        trs[1].dis_adjs = [None, None]
        inject_sampler(trs[1], ([ho_lab, he_lab], [ho_idx, he_idx]))
        out, logs, cls_logs = {}, [], []
        for ep in range(3):
            lg = ct.train_step((x, adj), labels, ep)              # main.py:313-329: the fine-tuning step comes first,
            cls_logs.append([lg[k] for k in ("loss_train", "acc_train", "loss_reg", "loss_val", "acc_val")])
            logs.append(trs[0].train_step((x, adj), None)["loss_heads_sup"])      # main.py:335-352: then every SSL trainer
            logs.append(trs[1].train_step((x, adj))["loss_head_disen"])
            logs.append(trs[2].train_step((x, adj))["loss_head_diversity"])
        lt = ct.test((x, adj), labels)
        out["cls_test"] = np.asarray([lt["loss_test"], lt["acc_test"], lt["roc_test"], lt["macroF_test"]], dtype=np.float64)
        out["cls_logs"] = np.asarray(cls_logs, dtype=np.float64)
        for nm, idxs in (("idx_train", ct.idx_train), ("idx_val", ct.idx_val), ("idx_test", ct.idx_test)):
            out[nm] = idxs.numpy().astype(np.int64)
        for nm in ("fuse1", "fuse2", "classifier"):
            for k, v in getattr(ct, nm).state_dict().items():
                out[f"cls.{nm}.{k}"] = np32(v)
        out["losses"] = np.asarray(logs, dtype=np.float32)
        for k, v in enc.state_dict().items():
            out["enc." + k] = np32(v)
        for t, tr in enumerate(trs):
            for nm in ("fuse1", "fuse2"):
                for k, v in getattr(tr, nm).state_dict().items():
                    out[f"t{t}.{nm}.{k}"] = np32(v)
        for nm in ("classifier1", "classifier2"):
            for k, v in getattr(trs[2], nm).state_dict().items():
                out[f"t2.{nm}.{k}"] = np32(v)
        np.savez_compressed(os.path.join(GOLD, f"tiny_traj_{gnn}_att{att}.npz"), **out)
        print("trajectory", gnn, att, "losses", np.round(out["losses"], 5).tolist())


def pack_param(out, key, v):
    """Real-graph trajectory fixtures stay small: the first 256 rows of a parameter + its sum and abs-sum (float64)."""
    a = np32(v)
    out[key] = a[:256] if a.ndim else a
    out[key + "#sum"] = np.float64(a.astype(np.float64).sum())
    out[key + "#abs"] = np.float64(np.abs(a.astype(np.float64)).sum())


def gen_trajectory_real():
    """The tiny-graph trajectory above at the real configuration (BASELINE configs[0] / [1]): Cora (bundled adjacency +
    the seeded surrogate features) and chameleon (real features), H = 8, nhid = 64, two epochs of ClsTrainer step ->
    SupEdge -> DisEdge -> DifHead (main.py:313-352's order) with the reference's trainers (pretrainer.py:709-763, 578-641,
    810-847; trainer.py:178-223) and injected pair lists, dropout 0."""
    for name, combos in (("cora", (("AT", 3), ("SAGE", 1))), ("chameleon", (("AT", 3), ("SAGE", 1), ("SAGE", 3)))):
        gen_trajectory_on(name, combos)


def gen_trajectory_on(name, combos):
    import random
    import trainer as ref_trainer
    idx, labels, feat, n = load_real(name)
    ei = torch.from_numpy(idx)
    lab = torch.from_numpy(labels)
    x = torch.from_numpy(feat) if feat is not None else ic.features(51, n, 64, "cora_surrogate")
    adj = sparse_adj(ei, torch.ones(ei.shape[1]), n)
    pos, homo, het = ic.edge_sets(ei, lab, n)
    sup_idx, sup_lab = ic.sample_pairs(61, n, pos, "sup")
    ho_idx, ho_lab = ic.sample_pairs(62, n, homo, "homo")
    he_idx, he_lab = ic.sample_pairs(63, n, het, "het")
    # conditioning probe: the same run on inputs perturbed by ~2 ulp.  Adam divides by sqrt(v): an element whose gradient is
    # a cancelling sum at fp32 noise level takes a +-lr step of EITHER sign, in the reference itself - how far each
    # parameter moves under that perturbation is recorded (`#sens`) and bounds what any fp32 implementation can be held to.
    pert = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).standard_normal(tuple(x.shape)).astype(np.float32))
    x_pert = x * (1.0 + 2.4e-7 * pert)
    for gnn, att in combos:
        seed = 300 + att

        def run_once(xin, full=None):
            a = ref_args(gnn, att, 8, 64, 64, extra=("--reg",) if att != 1 else ())
            a.lr, a.weight_decay = 0.01, 5e-4
            enc = ic.load_params(ref_models.DISGAT(a, nfeat=64, nhid=64, nclass=64, nheads=8, dropout=0.0), seed)
            random.seed(5)
            ct = ref_trainer.ClsTrainer(a, enc, lab, 1.5)
            ic.load_params(ct.fuse1, seed + 41)
            ic.load_params(ct.fuse2, seed + 42)
            ic.load_params(ct.classifier, seed + 43)
            trs = []
            for k, cls in enumerate((ref_pre.SupEdgeTrainer, ref_pre.GeneratedEdgeTrainer, ref_pre.DifHeadTrainer)):
                tr = cls(a, enc, [1.0, 0.5, 2.0][k])
                ic.load_params(tr.fuse1, seed + 1 + 10 * k)
                ic.load_params(tr.fuse2, seed + 2 + 10 * k)
                trs.append(tr)
            ic.load_params(trs[2].classifier1, seed + 4)
            ic.load_params(trs[2].classifier2, seed + 5)
            inject_sampler(trs[0], (sup_lab, [sup_idx]))
            trs[1].dis_adjs = [None, None]
            inject_sampler(trs[1], ([ho_lab, he_lab], [ho_idx, he_idx]))
            logs, cls_logs = [], []
            for ep in range(2):
                lg = ct.train_step((xin, adj), lab, ep)
                cls_logs.append([lg[k] for k in ("loss_train", "acc_train", "loss_reg", "loss_val", "acc_val")])
                logs.append(trs[0].train_step((xin, adj), None)["loss_heads_sup"])
                logs.append(trs[1].train_step((xin, adj))["loss_head_disen"])
                logs.append(trs[2].train_step((xin, adj))["loss_head_diversity"])
            params = {}
            for nm in ("fuse1", "fuse2", "classifier"):
                for k, v in getattr(ct, nm).state_dict().items():
                    params[f"cls.{nm}.{k}"] = v.detach().clone()
            for k, v in enc.state_dict().items():
                params["enc." + k] = v.detach().clone()
            for t, tr in enumerate(trs):
                for nm in ("fuse1", "fuse2"):
                    for k, v in getattr(tr, nm).state_dict().items():
                        params[f"t{t}.{nm}.{k}"] = v.detach().clone()
            for nm in ("classifier1", "classifier2"):
                for k, v in getattr(trs[2], nm).state_dict().items():
                    params[f"t2.{nm}.{k}"] = v.detach().clone()
            idxs = {nm: getattr(ct, nm).numpy().astype(np.int16) for nm in ("idx_train", "idx_val", "idx_test")}
            return params, np.asarray(logs, dtype=np.float64), np.asarray(cls_logs, dtype=np.float64), idxs

        params, logs, cls_logs, idxs = run_once(x)
        params_p, logs_p, cls_logs_p, _ = run_once(x_pert)
        out = {"cls_logs": cls_logs, "losses": logs.astype(np.float32), **idxs,
               "losses#sens": np.abs(logs - logs_p), "cls_logs#sens": np.abs(cls_logs - cls_logs_p)}
        worst = (0.0, "")
        for k, v in params.items():
            pack_param(out, k, v)
            dp = (v.double() - params_p[k].double()).abs()
            sens = float(dp.max())
            out[k + "#sens"] = np.float64(sens)
            out[k + "#nflip"] = np.int64(int((dp > 4e-4).sum()))      # elements the 2-ulp perturbation moves by more than PTOL
            if (name, gnn) == ("chameleon", "SAGE"):                  # (round 5; older fixtures stay byte-identical)
                out[k + "#med"] = np.float64(float(dp[:256].median()) if dp.dim() else float(dp))   # ... and the median movement of the kept rows
            worst = max(worst, (sens, k))
        np.savez_compressed(os.path.join(GOLD, f"{name}_traj_{gnn}_att{att}.npz"), **out)
        print(name, "trajectory", gnn, att, "losses", np.round(out["losses"], 5).tolist(), "cls", np.round(out["cls_logs"][-1], 5).tolist(),
              "| most perturbation-sensitive parameter", worst, "loss sens", out["losses#sens"].max())


def gen_labels():
    """GeneratedEdgeTrainer.get_label_all (pretrainer.py:386-513) on the tiny graph: the homo / hetero edge groups with every
    node label known (:448-456) and under --conformT (:465-498, train + val nodes of utils.split only), as flat row*N+col ids."""
    import random
    idx, vals, n = ic.tiny_graph()
    adj = sparse_adj(idx, vals, n)
    x = ic.features(21, n, 16)
    labels = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).integers(0, 3, n))
    out = {}
    for tag, extra in (("all", ()), ("conformT", ("--conformT",))):
        a = ref_args("AT", 3, 4, 16, 16, extra=extra)
        enc = ref_models.DISGAT(a, nfeat=16, nhid=16, nclass=16, nheads=4, dropout=0.0)
        tr = ref_pre.GeneratedEdgeTrainer(a, enc, 1.0)
        random.seed(11)
        groups = tr.get_label_all(x, adj, labels, load=False)
        for nm, grp in zip(("homo", "hetero"), groups):
            nz = grp.nonzero()
            out[f"{tag}.{nm}"] = (nz[:, 0] * n + nz[:, 1]).numpy().astype(np.int64)
    np.savez_compressed(os.path.join(GOLD, "tiny_edge_groups.npz"), **out)
    print("edge groups", {k: len(v) for k, v in out.items()})


# --------------------------------------------------------------------------- real graphs
def processed_index_set(a):
    """Index set of data_load.load_data's processed adjacency (data_load.py:66-81):
    nonzeros of (A + A^T + I).  a: scipy sparse [N,N]."""
    n = a.shape[0]
    b = ((a + a.T + sp.eye(n, format="csr")) != 0).tocoo()
    flat = np.unique(b.row.astype(np.int64) * n + b.col)
    return np.stack([flat // n, flat % n])


def load_real(name):
    d = os.path.join(REF, "data", name)
    labels = np.load(os.path.join(d, "label.npy")).astype(np.int64)
    if name == "chameleon":
        el = np.load(os.path.join(d, "adj_1.npy")).astype(np.int64)          # edge list (data_load.py:46-49)
        n = int(el.max()) + 1
        a = sp.coo_matrix((np.ones(len(el)), (el[:, 0], el[:, 1])), shape=(n, n)).tocsr()
        f = np.load(os.path.join(d, "feature_new.npy"))
        rs = f.sum(1)                                                         # data_load.py:137-144
        rinv = np.where(rs == 0, 0.0, 1.0 / np.where(rs == 0, 1.0, rs))
        feat = (f * rinv[:, None]).astype(np.float32)
    else:
        a = sp.load_npz(os.path.join(d, "adj_1_sp.npz")).tocsr()
        n = a.shape[0]
        feat = None                                                           # .MISSING_LARGE_BLOBS -> surrogate
    return processed_index_set(a), labels, feat, n


def gen_real(skip_existing=False):
    # chameleon: cross-check my O(E) preprocessing against the reference loader itself
    import data_load as ref_dl
    cwd = os.getcwd()
    a = ref_args("AT", 3, 8, 64, 64)
    a.origin_feat = False
    adjs, ref_feat, ref_lab = ref_dl.load_data(a, path=os.path.join(REF, "data/chameleon/"), dataset="chameleon", edge_type=1)
    ref_idx = adjs[0].coalesce().indices().numpy()
    idx, labels, feat, n = load_real("chameleon")
    assert np.array_equal(ref_idx, idx), "chameleon preprocessing mismatch"
    assert np.allclose(ref_feat.numpy(), feat, rtol=0, atol=0), "chameleon feature mismatch"
    assert np.array_equal(ref_lab.numpy(), labels)
    print("chameleon preprocessing == reference loader: N", n, "E", idx.shape[1])
    os.chdir(cwd)

    for name in ("cora", "chameleon", "cora_full"):
        idx, labels, feat, n = load_real(name)
        dt = np.int16 if n < 32768 else np.int32
        data = {"edge_index": idx.astype(dt), "labels": labels.astype(np.int16), "n": np.int64(n)}
        if feat is not None:
            data["features"] = feat
        np.savez_compressed(os.path.join(GOLD, f"data_{name}.npz"), **data)
        print("data", name, "N", n, "E", idx.shape[1], "classes", labels.max() + 1)

    for name, combos in (("cora", [(g, t) for g in GNNS for t in ATTS]),
                         ("chameleon", [(g, t) for g in GNNS for t in ATTS]),
                         ("cora_full", [(g, t) for g in GNNS for t in ATTS])):
        if skip_existing:
            combos = [(g, t) for g, t in combos if not os.path.exists(os.path.join(GOLD, f"{name}_{g}_att{t}.npz"))]
            if not combos:
                continue
        idx, labels, feat, n = load_real(name)
        ei = torch.from_numpy(idx)
        lab = torch.from_numpy(labels)
        x = torch.from_numpy(feat) if feat is not None else ic.features(51, n, 64, "cora_surrogate")
        adj = sparse_adj(ei, torch.ones(ei.shape[1]), n)
        pos, homo, het = ic.edge_sets(ei, lab, n)
        sup_idx, sup_lab = ic.sample_pairs(61, n, pos, "sup")
        ho_idx, ho_lab = ic.sample_pairs(62, n, homo, "homo")
        he_idx, he_lab = ic.sample_pairs(63, n, het, "het")
        for gnn, att in combos:
            full = {}
            run_case(x, adj, n, lab, gnn, att, 8, 64, 200 + att, [sup_idx],
                     (sup_lab, [sup_idx]), ([ho_lab, he_lab], [ho_idx, he_idx]), False, full)
            out = {k: full[k] for k in ("loss_sup", "loss_dis", "loss_dif")}
            for k in ("forward", "get_em_0", "get_em_1"):
                out[k + "_head"] = full[k][:256]
                out[k + "_colsum"] = full[k].astype(np.float64).sum(0)
                out[k + "_abssum"] = np.float64(np.abs(full[k].astype(np.float64)).sum())
            stride = max(1, ei.shape[1] // 2048)
            for l in range(2):
                out[f"adjs_{l}_sub"] = full[f"adjs_{l}"][:, ::stride]
                out[f"adjs_{l}_sum"] = full[f"adjs_{l}"].astype(np.float64).sum(1)
                out[f"aux_{l}_0_sub"] = full[f"aux_{l}_0"][:, :: max(1, sup_idx.shape[1] // 2048)]
                out[f"aux_{l}_0_sum"] = full[f"aux_{l}_0"].astype(np.float64).sum(1)
                out[f"edge_em_{l}_sum"] = full[f"edge_em_{l}"].astype(np.float64).sum((1, 2))
            np.savez_compressed(os.path.join(GOLD, f"{name}_{gnn}_att{att}.npz"), **out)
            print(name, gnn, att, "sum_em0 %.6e sum_em1 %.6e sup %.8f dis %.8f dif %.6f" % (
                full["get_em_0"].sum(), full["get_em_1"].sum(), out["loss_sup"], out["loss_dis"], out["loss_dif"]))


GRAD_REAL = (("cora", "AT", 3), ("cora", "SAGE", 1), ("chameleon", "AT", 3), ("chameleon", "SAGE", 1))
GRAD_REAL_HEADS = (0, 3, 7)


def gen_grads_real():
    """First-step parameter gradients on the real graphs: one loss.backward() per reference trainer
    (pretrainer.py:750-752, 629-631, 834-836) at H = 8, nhid = 64 with the fixtures' own pair lists - before Adam's sign
    amplification a gradient is well conditioned and can be pinned per element.  Kept: every parameter's sum and absolute
    sum; the full gradient of heads 0, 3, 7 of both layers and of every trainer-side parameter (fusers, classifiers)."""
    for name, gnn, att in GRAD_REAL:
        idx, labels, feat, n = load_real(name)
        ei = torch.from_numpy(idx)
        lab = torch.from_numpy(labels)
        x = torch.from_numpy(feat) if feat is not None else ic.features(51, n, 64, "cora_surrogate")
        adj = sparse_adj(ei, torch.ones(ei.shape[1]), n)
        pos, homo, het = ic.edge_sets(ei, lab, n)
        sup_idx, sup_lab = ic.sample_pairs(61, n, pos, "sup")
        ho_idx, ho_lab = ic.sample_pairs(62, n, homo, "homo")
        he_idx, he_lab = ic.sample_pairs(63, n, het, "het")
        full = {}
        run_case(x, adj, n, lab, gnn, att, 8, 64, 200 + att, [sup_idx], (sup_lab, [sup_idx]),
                 ([ho_lab, he_lab], [ho_idx, he_idx]), True, full)
        out = {k: full[k] for k in ("loss_sup", "loss_dis", "loss_dif")}
        for k, v in full.items():
            if not k.startswith("g") or "." not in k or k.split(".")[0] not in ("gsup", "gdis", "gdif"):
                continue
            out[k + "#sum"] = np.float64(v.astype(np.float64).sum())
            out[k + "#abs"] = np.float64(np.abs(v.astype(np.float64)).sum())
            part = k.split(".")[2] if k.split(".")[1] == "enc" else ""
            head = int(part.split("_")[1]) if part.startswith("attention") else None
            if (head is None and not part.startswith("fuser")) or head in GRAD_REAL_HEADS:
                out[k] = v
        np.savez_compressed(os.path.join(GOLD, f"{name}_grads_{gnn}_att{att}.npz"), **out)
        print("grads_real", name, gnn, att, "arrays", len(out), "sup %.8f dis %.8f dif %.6f" % (out["loss_sup"], out["loss_dis"], out["loss_dif"]))


def gen_rows():
    """Every ROW of the five entry points on one real graph (the real-graph fixtures keep 256-row slices,
    column sums and strided samples): chameleon (real features), H = 8, nhid = 64, AT att 3 (the example script's choice)
    and SAGE att 2 (the argparse default attention type) - per row the sum and the abs-sum over the features of forward /
    get_em, per head and 64-entry block the sums of the edge scores and of the aux scores, per head and row the sum of
    get_edge_em; plus the row maxima, the tolerance scale of a row.  All rows, ~200 KB per fixture."""
    name = "chameleon"
    idx, labels, feat, n = load_real(name)
    ei = torch.from_numpy(idx)
    lab = torch.from_numpy(labels)
    x = torch.from_numpy(feat)
    adj = sparse_adj(ei, torch.ones(ei.shape[1]), n)
    pos, homo, het = ic.edge_sets(ei, lab, n)
    sup_idx, sup_lab = ic.sample_pairs(61, n, pos, "sup")
    ho_idx, ho_lab = ic.sample_pairs(62, n, homo, "homo")
    he_idx, he_lab = ic.sample_pairs(63, n, het, "het")

    def blocks(a, b=64):                       # [H, M] -> [H, ceil(M / b)] sums (float64)
        a = a.astype(np.float64)
        pad = (-a.shape[1]) % b
        return np.pad(a, ((0, 0), (0, pad))).reshape(a.shape[0], -1, b).sum(-1)

    for gnn, att in (("AT", 3), ("SAGE", 2)):
        full = {}
        run_case(x, adj, n, lab, gnn, att, 8, 64, 200 + att, [sup_idx], (sup_lab, [sup_idx]),
                 ([ho_lab, he_lab], [ho_idx, he_idx]), False, full)
        out = {}
        for k in ("forward", "get_em_0", "get_em_1"):
            v = full[k].astype(np.float64)
            out[k + "_rowsum"], out[k + "_rowabs"], out[k + "_rowmax"] = v.sum(1), np.abs(v).sum(1), np.abs(v).max(1)
        for l in range(2):
            out[f"adjs_{l}_blk"] = blocks(full[f"adjs_{l}"])
            out[f"adjs_{l}_blkabs"] = blocks(np.abs(full[f"adjs_{l}"]))
            out[f"aux_{l}_0_blk"] = blocks(full[f"aux_{l}_0"])
            out[f"aux_{l}_0_blkabs"] = blocks(np.abs(full[f"aux_{l}_0"]))
            ee = full[f"edge_em_{l}"].astype(np.float64)                      # [H, N, F_in + nhid]
            out[f"edge_em_{l}_rowsum"], out[f"edge_em_{l}_rowabs"] = ee.sum(2), np.abs(ee).sum(2)
        np.savez_compressed(os.path.join(GOLD, f"{name}_rows_{gnn}_att{att}.npz"), **out)
        print(name, "rows", gnn, att, {k: v.shape for k, v in out.items() if k.endswith(("rowsum", "_blk"))})


TINY256 = dict(n=2048, e=40960, f=256, nhid=256, heads=4)


def gen_tiny256():
    """A fixture at a width the plane-operand GEMM chain tiles (nhid 64 on the bundled graphs never engages
    csrc/gemm_planes.hip, so the chain bench.py times had only met the float64 oracle): SURVEY 8(d)'s power-law generator
    at N = 2 048 / E = 40 960, F_in = nhid = 256, H = 4 - att 3 with AT, SAGE and (round 5) GCN, the reference's default
    attention type, att 2, with AT and SAGE, att 1 with GCN - the reference's five entry points and three losses; 256-row slices, column sums and
    strided score samples are kept.  Fixtures already on disk are left alone (they stay byte-identical)."""
    c = TINY256
    n = c["n"]
    idx = ic.powerlaw_index(1234, n, c["e"])
    ci = ic.coalesced_index_set(idx, n)
    adj = sparse_adj(idx, torch.ones(idx.shape[1]), n)
    x = ic.features(71, n, c["f"])
    labels = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).integers(0, 4, n))
    pos, homo, het = ic.edge_sets(ci, labels, n)
    sup_idx, sup_lab = ic.sample_pairs(81, n, pos, "sup")
    ho_idx, ho_lab = ic.sample_pairs(82, n, homo, "homo")
    he_idx, he_lab = ic.sample_pairs(83, n, het, "het")
    for gnn, att in (("AT", 3), ("SAGE", 3), ("GCN", 3), ("AT", 2), ("GCN", 1), ("SAGE", 2)):
        if os.path.exists(os.path.join(GOLD, f"tiny256_{gnn}_att{att}.npz")):
            continue
        full = {}
        run_case(x, adj, n, labels, gnn, att, c["heads"], c["nhid"], 400, [sup_idx], (sup_lab, [sup_idx]),
                 ([ho_lab, he_lab], [ho_idx, he_idx]), False, full)
        out = {k: full[k] for k in ("loss_sup", "loss_dis", "loss_dif")}
        for k in ("forward", "get_em_0", "get_em_1"):
            out[k + "_head"] = full[k][:256]
            out[k + "_colsum"] = full[k].astype(np.float64).sum(0)
            out[k + "_abssum"] = np.float64(np.abs(full[k].astype(np.float64)).sum())
        for l in range(2):
            out[f"adjs_{l}_sub"] = full[f"adjs_{l}"][:, :: max(1, ci.shape[1] // 2048)]
            out[f"adjs_{l}_sum"] = full[f"adjs_{l}"].astype(np.float64).sum(1)
            out[f"aux_{l}_0_sub"] = full[f"aux_{l}_0"][:, :: max(1, sup_idx.shape[1] // 2048)]
            out[f"aux_{l}_0_sum"] = full[f"aux_{l}_0"].astype(np.float64).sum(1)
            out[f"edge_em_{l}_head"] = full[f"edge_em_{l}"][:, :64]
            out[f"edge_em_{l}_sum"] = full[f"edge_em_{l}"].astype(np.float64).sum((1, 2))
        np.savez_compressed(os.path.join(GOLD, f"tiny256_{gnn}_att{att}.npz"), **out)
        print("tiny256", gnn, "att", att, "nnz", ci.shape[1], "M", sup_idx.shape[1], "sum_em0 %.6e sup %.8f dis %.8f dif %.6f" % (
            full["get_em_0"].sum(), out["loss_sup"], out["loss_dis"], out["loss_dif"]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--skip-existing", action="store_true", help="real graphs: only (graph, gnn, att) fixtures not on disk yet")
    o = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)      # the reference writes ./resource/... relative to CWD (pretrainer.py:390)
        if o.only in (None, "prims"):
            gen_prims()
        if o.only in (None, "tiny"):
            gen_tiny()
        if o.only in (None, "traj"):
            gen_trajectory()
        if o.only in (None, "traj", "traj_real"):
            gen_trajectory_real()
        if o.only in (None, "labels"):
            gen_labels()
        if o.only in (None, "real"):
            gen_real(o.skip_existing)
        if o.only in (None, "grads_real"):
            gen_grads_real()
        if o.only in (None, "tiny256"):
            gen_tiny256()
        if o.only in (None, "rows"):
            gen_rows()
        os.chdir(REPO)
