"""O(E) device-side pair sampling for the sparse SSL losses.

The reference builds dense N x N masks (pretrainer.py:683-707, 524-576):
    mask = (rand(N,N) < 3*rho) | first third of the shuffled positives;  indices = mask.nonzero().T
with rho = n_pos / N^2.  The same distribution is drawn here without the dense matrix:
Binomial(N^2, 3*rho) ~ 3*n_pos uniform flat indices (duplicates removed, as a mask would),
united with a random third of the positives, sorted row-major (= nonzero() order); labels are
membership in the positive set.
"""
import torch


def flat_edges(graph):
    """Sorted int64 row*N+col of every CSR entry."""
    return graph.row * graph.n + graph.col.to(torch.int64)


def sample_pairs(n, pos_flat, generator=None):
    """pos_flat: sorted unique int64 device tensor.  Returns (indices int64 [2,M], labels f32 [M])."""
    dev = pos_flat.device
    npos = int(pos_flat.shape[0])
    rand_flat = torch.randint(0, n * n, (3 * npos,), device=dev, generator=generator, dtype=torch.int64)
    sel = torch.randperm(npos, device=dev, generator=generator)[: npos // 3]
    flat = torch.unique(torch.cat([rand_flat, pos_flat[sel]]))
    labels = membership(flat, pos_flat)
    rows = torch.div(flat, n, rounding_mode="floor")
    return torch.stack([rows, flat - rows * n]), labels


def membership(flat, pos_flat):
    """float32 0/1: flat[i] in pos_flat (pos_flat sorted)."""
    if pos_flat.numel() == 0:
        return torch.zeros(flat.shape, dtype=torch.float32, device=flat.device)
    j = torch.searchsorted(pos_flat, flat).clamp_(max=pos_flat.numel() - 1)
    return (pos_flat[j] == flat).to(torch.float32)
