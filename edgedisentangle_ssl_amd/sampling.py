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


# ---------------------------------------------------------------------------------------------------------------
# Fixed-capacity form of the same sampler, for train_steps captured in a HIP graph (capture.py): every tensor has a shape
# that does not depend on the draw, nothing reads a size back to the host, and the list's valid length lives on the
# device.  The host contributes one number per step, K ~ Binomial(n_rows * n_cols, 3 rho) from its own generator (no device
# round trip), written into a device scalar before the graph is replayed.
#   draws   = `n_draw` iid uniform flat ids (n_draw covers K_max distinct ones with the expected duplicates and 2 % slack)
#   kept    = the first K DISTINCT values in draw order  (= a uniform K-subset, as uniform_subset() returns)
#   list    = sort(kept  U  a random third of the positives), duplicates merged; the tail up to the capacity is padding:
#             pair (n_rows-1, n_cols-1) - so the list stays row-major sorted - with label -1, which the pair-loss kernels
#             skip (include/disgat_hip.h: disgat_pair_loss) and which therefore carries a zero gradient through the backward.
class StaticSampler:
    def __init__(self, n_rows, pos_flat, n_cols=None, n_pos_global=None):
        import math
        self.n_rows, self.n_cols = int(n_rows), int(n_rows if n_cols is None else n_cols)
        self.pos = pos_flat
        self.npos = int(pos_flat.shape[0])
        n_glob = self.npos if n_pos_global is None else int(n_pos_global)
        self.total = self.n_rows * self.n_cols
        self.rho3 = min(1.0, 3.0 * n_glob / (float(self.n_cols) * float(self.n_cols)))
        mean = self.total * self.rho3
        self.k_max = int(min(self.total, math.ceil(mean + 8.0 * math.sqrt(max(mean * (1.0 - self.rho3), 1.0)) + 8)))
        if self.k_max >= self.total:
            raise ValueError("StaticSampler: the mask is (nearly) dense; use sample_pairs")
        # expected draws until k_max distinct ids have appeared: total * ln(total / (total - k_max))
        self.n_draw = int(math.ceil(self.total * math.log(self.total / (self.total - self.k_max)) * 1.02)) + 64
        self.capacity = self.k_max + self.npos // 3
        dev = pos_flat.device
        self.k_dev = torch.zeros((), dtype=torch.int64, device=dev)
        self.short = torch.zeros((), dtype=torch.int64, device=dev)        # > 0: some step found fewer than K distinct draws
        self.clamped = 0

    def draw_k(self, host_generator=None):
        """Host half of a step: K from the host generator into the device scalar the captured sampler reads."""
        k = binomial_count(self.total, self.rho3, host_generator)
        if k > self.k_max:                      # 8 sigma: never in practice; the list is then K_max long
            self.clamped += 1
            k = self.k_max
        self.k_dev.fill_(k)
        return k

    def sample(self, generator=None):
        """Device half (capturable): (indices int64 [2,C], labels f32 [C] with -1 padding, count f64 0-d)."""
        dev = self.pos.device
        total, nd = self.total, self.n_draw
        d = torch.randint(0, total, (nd,), device=dev, generator=generator, dtype=torch.int64)
        sv, si = torch.sort(d, stable=True)                               # ties stay in draw order
        first = torch.ones(nd, dtype=torch.bool, device=dev)
        first[1:] = sv[1:] != sv[:-1]                                     # first draw of each distinct value
        by_draw = torch.zeros(nd, dtype=torch.int64, device=dev).scatter_(0, si, first.to(torch.int64))
        rank = torch.cumsum(by_draw, 0)                                   # distinct values seen up to this draw
        keep = (by_draw > 0) & (rank <= self.k_dev)
        self.short += (rank[-1] < self.k_dev).to(torch.int64)
        rand_vals = torch.where(keep[si], sv, torch.full_like(sv, total))
        n3 = self.npos // 3
        sel = torch.argsort(torch.rand(self.npos, device=dev, generator=generator))[:n3]     # pretrainer.py:697-700
        cs = torch.sort(torch.cat([rand_vals, self.pos[sel]])).values
        dup = torch.zeros_like(cs, dtype=torch.bool)
        dup[1:] = cs[1:] == cs[:-1]
        flat = torch.sort(torch.where(dup, torch.full_like(cs, total), cs)).values[: self.capacity]
        valid = flat < total
        count = valid.sum().to(torch.float64)
        labels = torch.where(valid, membership(flat.clamp(max=total - 1), self.pos), torch.full((), -1.0, device=dev))
        flat = torch.where(valid, flat, torch.full_like(flat, total - 1))
        rows = torch.div(flat, self.n_cols, rounding_mode="floor")
        idx = torch.stack([rows, flat - rows * self.n_cols])
        idx._disgat_checked = (self.n_rows, self.n_cols, idx._version)
        idx._disgat_static = True          # ops_bwd: segment structures without host round trips
        labels._disgat_count = count       # pretrainer._pair_loss_value: the mean runs over the valid entries
        return idx, labels


class FixedList:
    """A given pair list behind StaticSampler's interface (tests and callers that bring their own lists to a captured step)."""

    def __init__(self, indices, labels, n_rows=None, n_cols=None):
        self.indices, self.labels = indices.contiguous(), labels.contiguous()
        if n_rows is not None:
            self.indices._disgat_checked = (int(n_rows), int(n_rows if n_cols is None else n_cols), self.indices._version)
        self.indices._disgat_static = True
        self.labels._disgat_count = torch.full((), float(labels.shape[0]), dtype=torch.float64, device=labels.device)

    def draw_k(self, host_generator=None):
        return int(self.labels.shape[0])

    def sample(self, generator=None):
        return self.indices, self.labels
