"""Pins oracle/disgat_oracle.py against fixtures produced by the unmodified
reference (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import inputs_common as ic
from oracle import disgat_oracle as orc

GNNS = ["AT", "SAGE", "GCN"]
ATTS = [1, 2, 3]


def close(a, b, rtol=2e-6, atol=None, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    tol = rtol * scale if atol is None else atol
    err = float(np.abs(a - b).max()) if b.size else 0.0
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert err <= tol, f"{what}: max|d|={err:.3e} tol={tol:.3e} scale={scale:.3e}"


def shapes_disgat(gnn, att, nfeat, nhid, nheads):
    s = {}
    for layer, fin in ((1, nfeat), (2, nhid)):
        for i in range(nheads):
            p = f"attention{layer}_{i}."
            if att == 3:
                s[p + "W"], s[p + "a"] = (2 * fin, nhid), (nhid, 1)
            else:
                s[p + "W"], s[p + "a"] = (fin, nhid), (2 * nhid, 1)
            if gnn == "AT":
                s[p + "W_em"] = (fin, nhid)
            elif gnn == "SAGE":
                s[p + "ag_layer.proj.weight"] = (nhid, 2 * fin)
            else:
                s[p + "ag_layer.weight"], s[p + "ag_layer.bias"] = (fin, nhid), (nhid,)
    # DISGAT's own (unused) fusers are part of its state_dict (models.py:174-179)
    for f in ("fuser1", "fuser2"):
        s[f + ".fuse.weight"], s[f + ".fuse.bias"] = (nhid, nhid * nheads), (nhid,)
    return s


def shapes_fuser(nhid, nheads):
    return {"fuse.weight": (nhid, nhid * nheads), "fuse.bias": (nhid,)}


def shapes_layer(gnn, att, fin, nhid):
    s = shapes_disgat(gnn, att, fin, nhid, 1)
    return {k[len("attention1_0."):]: v for k, v in s.items() if k.startswith("attention1_0.")}


def shapes_mlp(i, h, o):
    return {"model.0.weight": (h, i), "model.0.bias": (h,), "model.2.weight": (o, h), "model.2.bias": (o,)}


def fusers_from(seed, nhid, nheads):
    out = []
    for k in (1, 2):
        p = ic.make_params(shapes_fuser(nhid, nheads), seed + k)
        out.append(lambda heads, res, p=p: orc.fuse_layer(p, heads, res))
    return out


def test_prims(golden_dir):
    g = np.load(os.path.join(golden_dir, "prims.npz"))
    idx, _, n = ic.tiny_graph()
    ci = ic.coalesced_index_set(idx, n)
    v = torch.from_numpy(g["softmax_in"])
    sm = orc.sp_softmax(ci[0], v, n)
    close(sm, g["softmax_out"], what="sp_softmax")
    close(orc.sp_matmul(ci[0], ci[1], sm, torch.from_numpy(g["mat"])), g["matmul_out"], what="sp_matmul")
    rec, tgt = torch.from_numpy(g["mse_rec"]), torch.from_numpy(g["mse_tgt"])
    close(orc.adj_mse_loss(rec, tgt), g["mse_1d"], what="mse 1-D quirk")
    close(orc.adj_mse_loss(rec[:400].reshape(20, 20), tgt[:400].reshape(20, 20)), g["mse_2d"], what="mse 2-D")
    feats = [torch.from_numpy(g[f"fuse_in{k}"]) for k in range(4)]
    res = torch.from_numpy(g["fuse_res"])
    for rt in (0, 1, 2):
        for nr in (0, 1):
            for ur in (0, 1):
                rd = 10 if ur else 0
                if rt == 0:
                    sh = {"fuse.weight": (8, 32 + rd), "fuse.bias": (8,)}
                elif rt == 1:
                    sh = {"fuse.weight": (16, 32 + rd), "fuse.bias": (16,), "fuse2.weight": (8, 16), "fuse2.bias": (8,)}
                else:
                    sh = {"fuse.weight": (8, 32), "fuse.bias": (8,)}
                    if rd:
                        sh.update({"fuse2.weight": (8, rd), "fuse2.bias": (8,)})
                p = ic.make_params(sh, 70 + rt)
                got = orc.fuse_layer(p, feats, res, residue_type=rt, fuse_no_relu=bool(nr), residue_dim=rd)
                close(got, g[f"fuse_rt{rt}_nr{nr}_res{ur}"], what=f"fuse rt{rt} nr{nr} res{ur}")
    p = ic.make_params(shapes_mlp(8, 6, 4), 80)
    close(orc.mlp(p, feats[0]), g["mlp_raw"], what="mlp")
    close(orc.mlp(p, feats[0], cls=True), g["mlp_cls"], what="mlp cls")


def tiny_inputs():
    idx, vals, n = ic.tiny_graph()
    x = ic.features(21, n, 16)
    labels = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).integers(0, 3, n))
    ci = ic.coalesced_index_set(idx, n)
    pos, homo, het = ic.edge_sets(ci, labels, n)
    sup = ic.sample_pairs(31, n, pos, "sup")
    ho = ic.sample_pairs(32, n, homo, "homo")
    he = ic.sample_pairs(33, n, het, "het")
    aux = [ic.aux_pairs(41, n, 300, "a0"), ic.aux_pairs(42, n, 150, "a1")]
    return x, ci, n, sup, ho, he, aux


@pytest.mark.parametrize("gnn", GNNS)
@pytest.mark.parametrize("att", ATTS)
def test_tiny_all_entry_points(golden_dir, gnn, att):
    g = np.load(os.path.join(golden_dir, f"tiny_{gnn}_att{att}.npz"))
    x, ci, n, sup, ho, he, aux = tiny_inputs()
    H, nhid, seed = 4, 16, 100 + att
    sd = ic.make_params(shapes_disgat(gnn, att, 16, nhid, H), seed)
    fus = fusers_from(seed, nhid, H)
    r = orc.disgat_pass(sd, x, ci, fus, H, att, gnn, aux)
    close(torch.log_softmax(r["feat"][1], 1), g["forward"], what="forward")
    for l in range(2):
        close(r["feat"][l], g[f"get_em_{l}"], what=f"get_em {l}")
        close(torch.stack([e[:, 0] for e in r["adjs"][l]]), g[f"adjs_{l}"], what=f"adjs {l}")
        for j in range(2):
            close(torch.stack([h[j][:, 0] for h in r["aux"][l]]), g[f"aux_{l}_{j}"], what=f"aux {l} {j}")
        close(torch.stack(r["edge_em"][l]), g[f"edge_em_{l}"], what=f"edge_em {l}")
    lp = ic.make_params(shapes_layer(gnn, att, 16, nhid), seed + 3)
    h, e, au = orc.disga_layer(x, ci, lp, att, gnn, aux)
    close(h, g["layer_h"], what="layer h")
    close(e[:, 0], g["layer_e"], what="layer e")
    for j in range(2):
        close(au[j][:, 0], g[f"layer_aux_{j}"], what="layer aux")


@pytest.mark.parametrize("gnn", GNNS)
@pytest.mark.parametrize("att", ATTS)
def test_tiny_losses_and_grads(golden_dir, gnn, att):
    g = np.load(os.path.join(golden_dir, f"tiny_{gnn}_att{att}.npz"))
    x, ci, n, sup, ho, he, aux = tiny_inputs()
    H, nhid, seed = 4, 16, 100 + att

    def fresh():
        sd = ic.make_params(shapes_disgat(gnn, att, 16, nhid, H), seed)
        for v in sd.values():
            v.requires_grad_(True)
        return sd

    def check_grads(prefix, sd, skip=("fuser1", "fuser2")):
        for k, v in sd.items():
            if k.startswith(skip):
                continue
            ref = g[f"{prefix}.{k}"]
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            close(got, ref, rtol=2e-5, what=f"{prefix}.{k}")

    sd = fresh()
    r = orc.disgat_pass(sd, x, ci, fusers_from(seed, nhid, H), H, att, gnn, [sup[0]])
    loss = orc.sup_edge_loss(r["aux"], sup[1])
    close(loss.detach(), g["loss_sup"], what="loss_sup")
    loss.backward()
    check_grads("gsup.enc", sd)

    sd = fresh()
    r = orc.disgat_pass(sd, x, ci, fusers_from(seed, nhid, H), H, att, gnn, [ho[0], he[0]])
    loss = orc.dis_edge_loss(r["aux"], ho[1], he[1])
    close(loss.detach(), g["loss_dis"], what="loss_dis")
    loss.backward()
    check_grads("gdis.enc", sd)

    sd = fresh()
    r = orc.disgat_pass(sd, x, ci, fusers_from(seed, nhid, H), H, att, gnn)
    c1 = ic.make_params(shapes_mlp(nhid + 16, nhid, H), seed + 4)
    c2 = ic.make_params(shapes_mlp(2 * nhid, nhid, H), seed + 5)
    loss = orc.dif_head_loss(r["edge_em"], c1, c2)
    close(loss.detach(), g["loss_dif"], rtol=5e-6, what="loss_dif")
    loss.backward()
    check_grads("gdif.enc", sd)


def test_reference_sampler_properties(golden_dir):
    """The reference's own sampler output (recorded) has the structure our O(E)
    sampler reproduces: row-major sorted unique pairs, labels = edge membership,
    at least a third of the positives present."""
    g = np.load(os.path.join(golden_dir, "tiny_ref_sampler.npz"))
    idx, _, n = ic.tiny_graph()
    ci = ic.coalesced_index_set(idx, n)
    pos = set((ci[0] * n + ci[1]).tolist())
    flat = g["sup_idx"][0].astype(np.int64) * n + g["sup_idx"][1]
    assert np.all(np.diff(flat) > 0)
    lab = np.array([f in pos for f in flat.tolist()], dtype=np.float32)
    assert np.array_equal(lab, g["sup_lab"])
    assert lab.sum() >= len(pos) // 3
    mine_idx, mine_lab = ic.sample_pairs(31, n, np.array(sorted(pos)), "sup")
    mflat = (mine_idx[0] * n + mine_idx[1]).numpy()
    assert np.all(np.diff(mflat) > 0)
    assert mine_lab.sum() >= len(pos) // 3
    assert 0.5 < len(mflat) / len(flat) < 2.0


@pytest.mark.parametrize("gnn,att", [("AT", 3), ("SAGE", 1)])
def test_chameleon_losses(golden_dir, gnn, att):
    """The oracle's three SSL losses on a REAL graph (chameleon, real features, H = 8, nhid = 64) against the values
    the unmodified reference's train_step logged (pretrainer.py:612-627, 727-739, 819-832)."""
    g = np.load(os.path.join(golden_dir, f"chameleon_{gnn}_att{att}.npz"))
    d = np.load(os.path.join(golden_dir, "data_chameleon.npz"))
    n = int(d["n"])
    ei = torch.from_numpy(d["edge_index"].astype(np.int64))
    lab = torch.from_numpy(d["labels"].astype(np.int64))
    x = torch.from_numpy(d["features"])
    pos, homo, het = ic.edge_sets(ei, lab, n)
    sup = ic.sample_pairs(61, n, pos, "sup")
    ho = ic.sample_pairs(62, n, homo, "homo")
    he = ic.sample_pairs(63, n, het, "het")
    H, nhid, seed = 8, 64, 200 + att
    sd = ic.make_params(shapes_disgat(gnn, att, x.shape[1], nhid, H), seed)
    with torch.no_grad():
        r = orc.disgat_pass(sd, x, ei, fusers_from(seed, nhid, H), H, att, gnn, [sup[0]])
        close(orc.sup_edge_loss(r["aux"], sup[1]), g["loss_sup"], rtol=1e-5, what="loss_sup")
        r = orc.disgat_pass(sd, x, ei, fusers_from(seed, nhid, H), H, att, gnn, [ho[0], he[0]])
        close(orc.dis_edge_loss(r["aux"], ho[1], he[1]), g["loss_dis"], rtol=1e-5, what="loss_dis")
        c1 = ic.make_params(shapes_mlp(nhid + x.shape[1], nhid, H), seed + 4)
        c2 = ic.make_params(shapes_mlp(2 * nhid, nhid, H), seed + 5)
        close(orc.dif_head_loss(r["edge_em"], c1, c2), g["loss_dif"], rtol=1e-5, what="loss_dif")
