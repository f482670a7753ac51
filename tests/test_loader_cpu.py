"""O(E) data loader against the fixtures produced by the reference's own load_data (CPU)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

REF_DATA = "/root/reference/data"


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference data directory only exists in the build container")
def test_load_data_matches_reference_processed_graph(golden_dir):
    from edgedisentangle_ssl_amd import data_load
    args = SimpleNamespace(origin_feat=False, hetero=True, used_edge=1, sparse=True)
    adjs, feats, labels = data_load.load_data(args, path=REF_DATA + "/chameleon/", dataset="chameleon", edge_type=1)
    d = np.load(os.path.join(golden_dir, "data_chameleon.npz"))      # written from the reference loader's output
    a = adjs[0].coalesce()
    assert np.array_equal(a.indices().numpy(), d["edge_index"].astype(np.int64))
    assert np.array_equal(feats.numpy(), d["features"])
    assert np.array_equal(labels.numpy(), d["labels"].astype(np.int64))
    rowsum = torch.zeros(a.shape[0]).index_add_(0, a.indices()[0], a.values())
    assert torch.allclose(rowsum, torch.ones_like(rowsum), atol=1e-5)            # row-normalised like data_load.py:73


def test_fixture_loader(golden_dir):
    from edgedisentangle_ssl_amd import data_load
    adj, feats, labels = data_load.load_fixture(os.path.join(golden_dir, "data_cora.npz"))
    assert adj.shape == (2708, 2708) and feats is None and labels.shape == (2708,)
    assert adj._nnz() == 13264


def test_partition_on_load_equals_sharding_the_full_graph(golden_dir, tmp_path):
    """load_data(..., rank, world) builds each rank's DistGraph straight from its slice of the processed
    host CSR; it must equal cutting the fully loaded graph (parallel.DistGraph.shard), cover every row and edge once,
    and hand every rank its own feature rows.  The on-disk files are written here in the reference's format
    (data_load.py:22-60: adj_1.npy edge list, feature_new.npy, label.npy) from the chameleon fixture."""
    from edgedisentangle_ssl_amd import data_load, parallel
    from edgedisentangle_ssl_amd.graph import CSRGraph
    d = np.load(os.path.join(golden_dir, "data_chameleon.npz"))
    n = int(d["n"])
    ei = d["edge_index"].astype(np.int64)
    raw = ei[:, ei[0] < ei[1]].T                          # one direction, no self loops: the loader re-symmetrises
    np.save(tmp_path / "adj_1.npy", raw)
    np.save(tmp_path / "label.npy", d["labels"])
    rng = np.random.Generator(np.random.PCG64(1))
    np.save(tmp_path / "feature_new.npy", rng.random((n, 12)) + 0.1)
    args = SimpleNamespace(origin_feat=False, hetero=True, used_edge=1, sparse=True)
    adjs, feats, labels = data_load.load_data(args, path=str(tmp_path) + "/", dataset="x", edge_type=1)
    full = CSRGraph.from_adj(adjs[0])
    assert np.array_equal(full.indices().numpy(), ei)
    world, rows, nnz = 3, 0, 0
    for rank in range(world):
        (dg,), f_loc, lab = data_load.load_data(args, path=str(tmp_path) + "/", dataset="x", edge_type=1, rank=rank, world=world)
        ref = parallel.DistGraph.shard(full, rank, world)
        assert (dg.n, dg.row_start, dg.n_global, dg.counts) == (ref.n, ref.row_start, ref.n_global, ref.counts)
        assert torch.equal(dg.rowptr, ref.rowptr) and torch.equal(dg.col, ref.col) and torch.equal(dg.row, ref.row)
        assert torch.equal(f_loc, feats[dg.row_start: dg.row_start + dg.n]) and torch.equal(lab, labels)
        rows += dg.n
        nnz += dg.nnz
    assert rows == n and nnz == full.nnz
