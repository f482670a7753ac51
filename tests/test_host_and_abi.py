"""CPU-only checks of the host logic and of the C-ABI library (load + exported symbols; no
compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import inputs_common as ic


def test_library_builds_loads_and_exports_header_symbols():
    from edgedisentangle_ssl_amd import _lib
    path = _lib.build()
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "disgat_hip.h")).read()
    declared = sorted(set(re.findall(r"^\s*(?:int|const char\*)\s+(disgat_\w+)\s*\(", hdr, flags=re.M)))
    assert declared, "no declarations parsed from the header"
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/disgat_hip.h but not exported"
    assert declared == _lib.exported_symbols(), (declared, _lib.exported_symbols())
    lib.disgat_abi_version.restype = ctypes.c_int
    from edgedisentangle_ssl_amd import _lib as binding
    assert lib.disgat_abi_version() == binding.ABI_VERSION      # the loader rebuilds / refuses a library of another ABI
    # a diagnostic build (-DRS_DIAG / -DBB_DIAG / -DDISGAT_PL_DIAG: stamps, ablation switches) must not be picked up silently:
    # the library reports the -D flags it was compiled with; _lib.load() rebuilds on a mismatch with this process's flags
    lib.disgat_build_flags.restype = ctypes.c_char_p
    assert lib.disgat_build_flags().decode() == binding._extra_flags() == ""


def test_a_library_built_with_other_flags_counts_as_stale(monkeypatch):
    """tools/*_ablate.sh build the library with -D... diagnostics; a later process with other DISGAT_HIPCC_FLAGS (normally
    none) must rebuild BEFORE its first dlopen - a loaded library cannot be swapped - which _lib decides from the flags file
    written next to the .so; the library itself reports the same string (checked after loading)."""
    from edgedisentangle_ssl_amd import _lib
    _lib.build()
    assert not _lib._stale() and open(_lib.FLAGS_PATH).read() == ""
    monkeypatch.setenv("DISGAT_HIPCC_FLAGS", "  -DRS_DIAG=1   -DFOO ")
    assert _lib._extra_flags() == "-DRS_DIAG=1 -DFOO" and _lib._stale()
    monkeypatch.delenv("DISGAT_HIPCC_FLAGS")
    assert not _lib._stale()
    saved = open(_lib.FLAGS_PATH).read()
    try:
        with open(_lib.FLAGS_PATH, "w") as f:
            f.write("-DBB_DIAG=32")                 # what a diagnostic build leaves behind
        assert _lib._stale()
        os.remove(_lib.FLAGS_PATH)                  # an installed library of unknown origin
        assert _lib._stale()
    finally:
        with open(_lib.FLAGS_PATH, "w") as f:
            f.write(saved)
    assert not _lib._stale()


def test_csr_and_work_items_cover_every_edge_once():
    from edgedisentangle_ssl_amd.graph import CSRGraph
    idx, vals, n = ic.tiny_graph()
    g = CSRGraph.from_index(idx, n)
    ci = ic.coalesced_index_set(idx, n)
    assert torch.equal(g.indices(), ci)                      # CSR order == coalesce() order
    assert int(g.rowptr[-1]) == g.nnz == ci.shape[1]
    for chunk in (4, 8, 64, 1024):
        wi = g.work_items(chunk)
        it = wi.items.long()
        assert it.shape == (wi.n_items, 4)
        lens = it[:, 2] - it[:, 1]
        assert int(lens.max()) <= chunk and int(lens.sum()) == g.nnz
        assert torch.all(lens[:-1] >= lens[1:])              # descending length
        cover = torch.zeros(g.nnz, dtype=torch.int64)
        for r, b, e, s in it.tolist():
            cover[b:e] += 1
            assert int(g.rowptr[r]) <= b <= e <= int(g.rowptr[r + 1])
        assert torch.all(cover == 1)
        rows_seen = torch.bincount(it[:, 0], minlength=n)
        assert torch.all(rows_seen >= 1)                     # empty rows keep one (empty) item
        split = it[it[:, 3] >= 0]
        assert sorted(split[:, 3].tolist()) == list(range(wi.n_slots))
        sp = wi.split_ptr.long()
        for k, r in enumerate(wi.split_rows.tolist()):
            slots = sorted(split[split[:, 0] == r][:, 3].tolist())
            assert slots == list(range(int(sp[k]), int(sp[k + 1])))


def test_transpose_structure():
    from edgedisentangle_ssl_amd.graph import CSRGraph
    idx, vals, n = ic.tiny_graph()
    g = CSRGraph.from_index(idx, n)
    t = g.transpose()
    # entry j of the transpose: column c (CSC order), source row t.col[j], forward edge id t.eid[j]
    eid = t.eid.long()
    assert torch.equal(g.col.long()[eid], t.row)
    assert torch.equal(g.row[eid], t.col.long())
    assert sorted(eid.tolist()) == list(range(g.nnz))


def test_sampler_matches_reference_structure(golden_dir):
    """The sampler's CPU restatement (the product is a HIP kernel held to equality with it, tests/test_gpu_sampler.py)."""
    from edgedisentangle_ssl_amd import sampling
    from edgedisentangle_ssl_amd.graph import CSRGraph
    from oracle import sampler_oracle as so
    idx, vals, n = ic.tiny_graph()
    g = CSRGraph.from_index(idx, n)
    pos = sampling.flat_edges(g)
    p = 3.0 * g.nnz / (n * n)
    items = sampling.build_items(pos, n, n, p).numpy()
    r, c, lab, _over = so.sample(items, g.col.numpy(), g.nnz // 3, p, 3, 0, n)
    pidx, lab = torch.from_numpy(np.stack([r, c])), torch.from_numpy(lab)
    flat = pidx[0] * n + pidx[1]
    assert torch.all(flat[1:] > flat[:-1])                   # row-major, unique (mask.nonzero() order)
    assert torch.equal(lab, torch.isin(flat, pos).float())
    assert int(lab.sum()) >= g.nnz // 3
    ref = np.load(os.path.join(golden_dir, "tiny_ref_sampler.npz"))
    assert 0.5 < pidx.shape[1] / ref["sup_idx"].shape[1] < 2.0


def test_adj_mse_loss_quirk(golden_dir):
    from edgedisentangle_ssl_amd.utils import adj_mse_loss
    g = np.load(os.path.join(golden_dir, "prims.npz"))
    rec, tgt = torch.from_numpy(g["mse_rec"]), torch.from_numpy(g["mse_tgt"])
    assert abs(float(adj_mse_loss(rec, tgt)) - float(g["mse_1d"])) < 1e-7
    assert abs(float(adj_mse_loss(rec[:400].reshape(20, 20), tgt[:400].reshape(20, 20))) - float(g["mse_2d"])) < 1e-7


def test_parser_surface():
    from edgedisentangle_ssl_amd.utils import get_parser
    a = get_parser().parse_args("--model=DISGAT --gnn_type SAGE --att 3 --nhead 8 --sparse --pretrain SupEdge DisEdge "
                                "DifHead --pre_weight 100 1 0 --pre_edge 1 1 1 --downstream CLS --down_weight 1.0 "
                                "--finetune --constrain_layer 0 --steps 5 --dataset cora_full".split())
    assert a.model == "DISGAT" and a.gnn_type == "SAGE" and a.att == 3 and a.nhead == 8 and a.sparse
    assert a.pretrain == ["SupEdge", "DisEdge", "DifHead"] and a.pre_weight == [100.0, 1.0, 0.0]
    d = get_parser().parse_args([])
    assert (d.att, d.nhead, d.nhid, d.dropout, d.gnn_type, d.seed, d.lr) == (2, 4, 64, 0.1, "AT", 4, 0.01)


def test_state_dict_keys_match_reference_layout():
    from types import SimpleNamespace
    from edgedisentangle_ssl_amd import DISGAT
    from test_oracle_golden import shapes_disgat
    for gnn in ("AT", "SAGE", "GCN"):
        for att in (1, 2, 3):
            a = SimpleNamespace(gnn_type=gnn, att=att, residue=False, residue_type=0, fuse_no_relu=False)
            m = DISGAT(a, nfeat=12, nhid=8, nclass=8, nheads=3, dropout=0.0)
            got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            assert got == shapes_disgat(gnn, att, 12, 8, 3)


def test_cpu_tensors_are_refused():
    from types import SimpleNamespace
    import edgedisentangle_ssl_amd as pkg
    a = SimpleNamespace(gnn_type="AT", att=3, residue=False, residue_type=0, fuse_no_relu=False)
    enc = pkg.DISGAT(a, nfeat=16, nhid=16, nclass=16, nheads=4, dropout=0.0)
    idx, vals, n = ic.tiny_graph()
    adj = torch.sparse_coo_tensor(idx, vals, (n, n))
    fus = [pkg.FuseLayer(a, 4, nfeat=16), pkg.FuseLayer(a, 4, nfeat=16)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc.get_em(ic.features(21, n, 16), adj, fus)


def test_graph_cache_is_per_tensor_object():
    from edgedisentangle_ssl_amd.graph import graph_of, _GRAPH_CACHE
    import gc
    idx, vals, n = ic.tiny_graph()
    a1 = torch.sparse_coo_tensor(idx, vals, (n, n))
    g1 = graph_of(a1)
    assert graph_of(a1) is g1                                 # same object -> cached
    a2 = torch.sparse_coo_tensor(idx[:, :50], vals[:50], (n, n))
    assert graph_of(a2) is not g1 and graph_of(a2).nnz != g1.nnz
    key = id(a1)
    del a1
    gc.collect()
    assert key not in _GRAPH_CACHE                            # entry dies with the tensor


def test_fuse_layer_and_mlp_variants_match_reference_goldens(golden_dir):
    """The product's FuseLayer (all residue_type / fuse_no_relu / residue combinations) and MLP against
    the reference's outputs (prims.npz); on CPU tensors they run their plain torch path."""
    import os
    from types import SimpleNamespace
    from edgedisentangle_ssl_amd import FuseLayer, MLP
    g = np.load(os.path.join(golden_dir, "prims.npz"))
    feats = [torch.from_numpy(g[f"fuse_in{k}"]) for k in range(4)]
    res = torch.from_numpy(g["fuse_res"])
    for rt in (0, 1, 2):
        for nr in (0, 1):
            for ur in (0, 1):
                a = SimpleNamespace(residue_type=rt, fuse_no_relu=bool(nr))
                fl = ic.load_params(FuseLayer(a, 4, nfeat=8, residue=10 if ur else 0), 70 + rt)
                with torch.no_grad():
                    got = fl(feats, res)
                assert np.abs(got.numpy() - g[f"fuse_rt{rt}_nr{nr}_res{ur}"]).max() < 2e-6, (rt, nr, ur)
    mlp = ic.load_params(MLP(in_feat=8, hidden_size=6, out_size=4, layers=2), 80)
    with torch.no_grad():
        assert np.abs(mlp(feats[0]).numpy() - g["mlp_raw"]).max() < 2e-6
        assert np.abs(mlp(feats[0], cls=True).numpy() - g["mlp_cls"]).max() < 2e-6


def test_constrain_layer_gating():
    """pretrainer.py:597, 728: 0 = both layers, 1 = second layer only, 2 = none (the reference then
    crashes on `None * weight`; so does the mirror)."""
    from edgedisentangle_ssl_amd.pretrainer import Trainer
    t = object.__new__(Trainer)
    for cl, want in ((0, [True, True]), (1, [False, True]), (2, [False, False])):
        t.constrain_layer = cl
        assert [t._layer_on(i) for i in range(2)] == want


def test_same_seed_gives_the_reference_initial_parameters(golden_dir):
    """Parameter creation order and initialisers match the reference (models.py:163-179,
    layers.py:319-337, 33-36, 79): torch.manual_seed(s) yields bit-identical initial state_dicts."""
    import os
    from types import SimpleNamespace
    from edgedisentangle_ssl_amd import DISGAT
    g = np.load(os.path.join(golden_dir, "init_seed4.npz"))
    for gnn in ("AT", "SAGE", "GCN"):
        for att in (1, 3):
            a = SimpleNamespace(gnn_type=gnn, att=att, residue=False, residue_type=0, fuse_no_relu=False)
            torch.manual_seed(4)
            m = DISGAT(a, nfeat=20, nhid=12, nclass=12, nheads=3, dropout=0.1)
            sd = m.state_dict()
            keys = [k[len(f"{gnn}_{att}."):] for k in g.files if k.startswith(f"{gnn}_{att}.")]
            assert keys == list(sd.keys())
            for k in keys:
                assert np.array_equal(sd[k].numpy(), g[f"{gnn}_{att}.{k}"]), (gnn, att, k)


def test_group_correlation_matches_reference(golden_dir):
    import os
    from edgedisentangle_ssl_amd.utils import group_correlation
    g = np.load(os.path.join(golden_dir, "prims.npz"))
    got = group_correlation(torch.from_numpy(g["corr_in"]))
    assert np.allclose(got.numpy(), g["corr_out"], atol=2e-6)


def test_checkpoint_round_trip_uses_the_reference_layout(tmp_path):
    """main.save_model / load_model (main.py:214-235): {'encoder': state_dict} under the reference's path scheme."""
    from types import SimpleNamespace
    from edgedisentangle_ssl_amd import DISGAT, main
    a = SimpleNamespace(gnn_type="AT", att=3, residue=False, residue_type=0, fuse_no_relu=False, dataset="cora", model="DISGAT",
                        used_edge=1, pre_weight=[1.0, 1.0], reg=False, pretrain=["SupEdge", "DifHead"], load=7)
    torch.manual_seed(1)
    m1 = DISGAT(a, nfeat=20, nhid=12, nclass=12, nheads=3, dropout=0.1)
    path = main.save_model(m1, a, 7, root=str(tmp_path))
    assert path.endswith("checkpoint/cora/DISGAT_used_edge1_weight[1.0, 1.0]_regFalse/pretrain_['SupEdge', 'DifHead']_7.pth")
    assert list(torch.load(path).keys()) == ["encoder"]
    torch.manual_seed(2)
    m2 = DISGAT(a, nfeat=20, nhid=12, nclass=12, nheads=3, dropout=0.1)
    main.load_model(m2, a, root=str(tmp_path))
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_conformT_restricts_disedge_groups_to_the_labelled_split():
    """--conformT (pretrainer.py:465-498): only edges whose two endpoints are in the train + val split count as known
    homo / hetero edges; without it every edge does (pretrainer.py:448-456).  Pure host logic: runs on the CPU."""
    import random
    from types import SimpleNamespace
    from edgedisentangle_ssl_amd import pretrainer, sampling
    from edgedisentangle_ssl_amd.graph import CSRGraph
    from edgedisentangle_ssl_amd.utils import split
    idx, _vals, n = ic.tiny_graph()
    g = CSRGraph.from_index(idx, n)
    labels = torch.arange(n) % 3
    tr = pretrainer.GeneratedEdgeTrainer.__new__(pretrainer.GeneratedEdgeTrainer)
    tr.dis_type = 1
    tr.args = SimpleNamespace(conformT=False, node_sup_ratio=0.25)
    homo, het = tr.get_label_all(None, g, labels)
    flat = sampling.flat_edges(g)
    same = labels[g.row] == labels[g.col.long()]
    assert torch.equal(homo, flat[same]) and torch.equal(het, flat[~same]) and homo.numel() + het.numel() == g.nnz
    tr.args.conformT = True
    random.seed(11)
    homo_c, het_c = tr.get_label_all(None, g, labels)
    random.seed(11)
    tr_i, va_i, _te, _m = split(labels, train_ratio=0.25)
    known = torch.zeros(n, dtype=torch.bool)
    known[torch.cat((tr_i, va_i))] = True
    for sub, full in ((homo_c, homo), (het_c, het)):
        assert torch.isin(sub, full).all() and sub.numel() < full.numel()
        assert known[sub // n].all() and known[sub % n].all()
    both = known[g.row] & known[g.col.long()]
    assert homo_c.numel() == int((same & both).sum()) and het_c.numel() == int((~same & both).sum())
    assert tr.n_pos_global == [homo_c.numel(), het_c.numel()]


def test_split_reproduces_the_reference_index_sets(golden_dir):
    """utils.split (utils.py:118-161 of the reference) draws with python's `random`: under the same seed the train / val /
    test node sets equal the ones the reference produced when the trajectory goldens were recorded."""
    import random
    from edgedisentangle_ssl_amd.utils import split
    g = np.load(os.path.join(golden_dir, "tiny_traj_AT_att3.npz"))
    labels = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).integers(0, 3, 64))
    random.seed(5)
    tr, va, te, mat = split(labels, train_ratio=0.25)
    np.testing.assert_array_equal(tr.numpy(), g["idx_train"])
    np.testing.assert_array_equal(va.numpy(), g["idx_val"])
    np.testing.assert_array_equal(te.numpy(), g["idx_test"])
    assert int(mat[:, 0].sum()) == len(tr) and int(mat[:, 2].sum()) == len(te)


@pytest.mark.parametrize("conform", [False, True])
def test_edge_groups_equal_the_reference_label_builder(golden_dir, conform):
    """GeneratedEdgeTrainer.get_label_all against the groups the reference built from its dense N x N masks
    (pretrainer.py:448-456 and, under --conformT, :465-498) on the tiny graph: same flat row*N+col id sets."""
    import random
    from types import SimpleNamespace
    from edgedisentangle_ssl_amd import pretrainer
    from edgedisentangle_ssl_amd.graph import CSRGraph
    gold = np.load(os.path.join(golden_dir, "tiny_edge_groups.npz"))
    idx, _vals, n = ic.tiny_graph()
    g = CSRGraph.from_index(idx, n)
    labels = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).integers(0, 3, n))
    tr = pretrainer.GeneratedEdgeTrainer.__new__(pretrainer.GeneratedEdgeTrainer)
    tr.dis_type = 1
    tr.args = SimpleNamespace(conformT=conform, node_sup_ratio=0.25)
    random.seed(11)
    homo, het = tr.get_label_all(None, g, labels)
    tag = "conformT" if conform else "all"
    np.testing.assert_array_equal(homo.numpy(), gold[f"{tag}.homo"])
    np.testing.assert_array_equal(het.numpy(), gold[f"{tag}.hetero"])
