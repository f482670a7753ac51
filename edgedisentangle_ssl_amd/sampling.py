"""O(E) device-side pair sampling for the sparse SSL losses.

The reference builds dense N x N masks (pretrainer.py:683-707, 524-576):
    mask = (rand(N,N) < 3*rho) | first third of the shuffled positives;  indices = mask.nonzero().T
with rho = n_pos / N^2.  The same distribution is drawn here without the dense matrix: the number of entries the
Bernoulli mask switches on is K ~ Binomial(N^2, 3*rho) and, given K, they are a uniform K-subset of the N^2
positions; that subset is drawn directly (uniform flat indices, duplicates redrawn until K distinct ones exist),
united with a random third of the positives and sorted row-major (= nonzero() order); labels are membership in the
positive set.

On a row shard (parallel.DistGraph) a rank samples over ITS rows only: flat ids are local_row * n_cols + global_col,
n_cols = the global node count, and rho stays the global density, so the union over ranks has the unsharded
distribution.
"""
import torch


def flat_edges(graph):
    """Sorted int64 row*n_cols+col of every CSR entry (row local, col global on a shard)."""
    return graph.row * graph.n_cols + graph.col.to(torch.int64)


def binomial_count(total, p, generator=None):
    """K ~ Binomial(total, p) for total up to ~1e14 (float64 on the host generator: one scalar)."""
    if total <= 0 or p <= 0:
        return 0
    p = min(float(p), 1.0)
    return int(torch.binomial(torch.tensor(float(total), dtype=torch.float64), torch.tensor(p, dtype=torch.float64),
                              generator=generator).item())


def uniform_subset(total, k, device, generator=None):
    """k distinct int64 ids uniform over [0,total), sorted.  Draw with replacement, drop duplicates, top up."""
    k = min(int(k), int(total))
    if k == 0:
        return torch.empty(0, dtype=torch.int64, device=device)
    have = torch.unique(torch.randint(0, total, (k,), device=device, generator=generator, dtype=torch.int64))
    while have.numel() < k:
        missing = k - have.numel()
        extra = torch.randint(0, total, (missing + (missing >> 3) + 8,), device=device, generator=generator, dtype=torch.int64)
        extra = torch.unique(extra)
        extra = extra[~membership(extra, have).bool()]
        if extra.numel() > missing:                 # keep a uniform subset of the fresh ids
            extra = extra[torch.randperm(extra.numel(), device=device, generator=generator)[:missing]]
        have = torch.sort(torch.cat([have, extra])).values
    return have


def sample_pairs(n_rows, pos_flat, generator=None, n_cols=None, n_pos_global=None, host_generator=None):
    """pos_flat: sorted unique int64 device tensor of this process's positives (row * n_cols + col).
    n_rows: rows this process owns; n_cols: number of nodes a column can name (default n_rows = unsharded);
    n_pos_global: positives over all ranks (default: len(pos_flat)).
    Returns (indices int64 [2,M], labels f32 [M]) - rows local, columns global, row-major sorted."""
    dev = pos_flat.device
    n_cols = n_rows if n_cols is None else n_cols
    npos = int(pos_flat.shape[0])
    n_glob = npos if n_pos_global is None else int(n_pos_global)
    rho3 = 3.0 * n_glob / (float(n_cols) * float(n_cols))           # pretrainer.py:691-692: edge_ratio * 3
    total = n_rows * n_cols
    k = binomial_count(total, rho3, host_generator)
    rand_flat = uniform_subset(total, k, dev, generator)
    sel = torch.randperm(npos, device=dev, generator=generator)[: npos // 3]      # pretrainer.py:697-700
    flat = torch.unique(torch.cat([rand_flat, pos_flat[sel]]))
    labels = membership(flat, pos_flat)
    rows = torch.div(flat, n_cols, rounding_mode="floor")
    idx = torch.stack([rows, flat - rows * n_cols])
    idx._disgat_checked = (int(n_rows), int(n_cols), idx._version)   # in range by construction (ops.check_pairs): no round trip
    return idx, labels


def membership(flat, pos_flat):
    """float32 0/1: flat[i] in pos_flat (pos_flat sorted)."""
    if pos_flat.numel() == 0:
        return torch.zeros(flat.shape, dtype=torch.float32, device=flat.device)
    j = torch.searchsorted(pos_flat, flat).clamp_(max=pos_flat.numel() - 1)
    return (pos_flat[j] == flat).to(torch.float32)
