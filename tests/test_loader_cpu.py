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
