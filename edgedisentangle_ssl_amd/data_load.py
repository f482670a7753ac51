"""O(E) loader for the reference's on-disk graph format (the hot path's input contract).

Reads what data_load.load_data reads (data_load.py:22-94): `label.npy`, `feature_new.npy`
(row-sum normalised, data_load.py:137-144) or `feature.npy` with --origin_feat, and
`adj_{k}.npy` (edge list or dense) / `adj_{k}_sp.npz`, and produces the same objects:
adjacency = torch sparse COO of A + A^T + I, symmetrised and row-normalised, float32
(data_load.py:66-81, 158-165), features float32, labels int64.  Unlike the reference it never
builds a dense N x N array (np.fill_diagonal / todense at data_load.py:45, 69), so graphs beyond
~1e5 nodes load.
"""
import os

import numpy as np
import scipy.sparse as sp
import torch


def _load_edges(path, k):
    npy = os.path.join(path, "adj_{}.npy".format(k))
    if os.path.exists(npy):
        e = np.load(npy)
        if e.ndim == 2 and e.shape[1] == 2 and e.shape[0] != 2:        # edge list (data_load.py:46-49)
            e = e.astype(np.int64)
            n = int(e.max()) + 1
            return sp.coo_matrix((np.ones(len(e), dtype=np.float32), (e[:, 0], e[:, 1])), shape=(n, n)).tocsr()
        return sp.csr_matrix(e)
    return sp.load_npz(os.path.join(path, "adj_{}_sp.npz".format(k))).tocsr()


def process_adj(a):
    """(A + A^T + I > 0), row-normalised (data_load.py:69-73).  The reference symmetrises by taking the
    larger of a_ij, a_ji of a 0/1 matrix, i.e. the union of the supports."""
    n = a.shape[0]
    b = ((a + a.T + sp.eye(n, format="csr")) != 0).astype(np.float32).tocsr()
    rs = np.asarray(b.sum(1)).ravel()
    b = sp.diags(1.0 / rs).dot(b).tocoo()
    idx = torch.from_numpy(np.vstack((b.row, b.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(b.data.astype(np.float32)), (n, n))


def normalize_features(f):
    rs = f.sum(1)
    inv = np.where(rs == 0, 0.0, 1.0 / np.where(rs == 0, 1.0, rs))
    return (f * inv[:, None]).astype(np.float32)


def processed_csr(a):
    """Host CSR (indptr int64, indices int32) of the index set of (A + A^T + I): all the DISGAT path reads of the
    processed adjacency (layers.py:344 takes adj.coalesce().indices(); the row-normalised values are ignored)."""
    n = a.shape[0]
    b = ((a + a.T + sp.eye(n, format="csr")) != 0).tocsr()
    b.sort_indices()
    return b.indptr.astype(np.int64), b.indices.astype(np.int32)


def partition_graph(indptr, indices, rank, world, device, group=None):
    """This rank's nnz-balanced row range of a host CSR as a parallel.DistGraph, built directly from the slice: the
    replicated structure never reaches the device ("partitioned on load for multi-GPU")."""
    from .parallel import DistGraph, balanced_row_ranges
    n = indptr.shape[0] - 1
    b = balanced_row_ranges(torch.from_numpy(indptr), world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    e0, e1 = int(indptr[lo]), int(indptr[hi])
    rp = torch.from_numpy((indptr[lo:hi + 1] - e0).astype(np.int32)).to(device)
    col = torch.from_numpy(np.ascontiguousarray(indices[e0:e1])).to(device)
    row = torch.repeat_interleave(torch.arange(hi - lo, device=device), (rp[1:] - rp[:-1]).long())
    return DistGraph(hi - lo, rp.contiguous(), col.contiguous(), row.contiguous(), n, lo, (b[1:] - b[:-1]).tolist(), group)


def load_data(args, path="data/dblp/", dataset="dblp", edge_type=3, rank=None, world=None, device="cpu", group=None):
    """Same signature and return convention as the reference: (adjs, features, labels); adjs is a list
    of sparse tensors when args.hetero else a single one.

    rank / world (not in the reference): partition on load for a multi-GPU run - adjs become this rank's
    parallel.DistGraph (local rows, global columns, nnz-balanced), features the rank's own rows (both on `device`),
    labels stay global (the trainers index them with global node ids)."""
    labels = torch.from_numpy(np.load(os.path.join(path, "label.npy")).astype(np.int64))
    if getattr(args, "origin_feat", False):
        feats = np.load(os.path.join(path, "feature.npy")).astype(np.float32)
    else:
        feats = normalize_features(np.load(os.path.join(path, "feature_new.npy")))
    edges = [_load_edges(path, k + 1) for k in range(edge_type)]
    if args.hetero:
        use = edges
    elif args.used_edge == 0:
        tot = edges[0]
        for e in edges[1:]:
            tot = tot + e
        use = [tot]
    else:
        use = [edges[args.used_edge - 1]]
    if not args.sparse:
        raise NotImplementedError("dense adjacency (no --sparse) is outside the HIP path")
    if world is not None and world > 1:
        adjs = [partition_graph(*processed_csr(a), rank, world, device, group) for a in use]
        lo = adjs[0].row_start
        feats = torch.from_numpy(np.ascontiguousarray(feats[lo: lo + adjs[0].n])).to(device)
        return (adjs, feats, labels) if args.hetero else (adjs[0], feats, labels)
    adjs = [process_adj(a) for a in use]
    feats = torch.from_numpy(feats)
    return (adjs, feats, labels) if args.hetero else (adjs[0], feats, labels)


def load_fixture(npz_path, surrogate_features=None):
    """tests/golden/data_<name>.npz (processed index set + labels [+ features])."""
    d = np.load(npz_path)
    n = int(d["n"])
    ei = torch.from_numpy(d["edge_index"].astype(np.int64))
    deg = torch.bincount(ei[0], minlength=n).float()
    adj = torch.sparse_coo_tensor(ei, 1.0 / deg[ei[0]], (n, n))
    feats = torch.from_numpy(d["features"]) if "features" in d.files else surrogate_features
    return adj, feats, torch.from_numpy(d["labels"].astype(np.int64))
