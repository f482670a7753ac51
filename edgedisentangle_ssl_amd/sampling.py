"""SSL training-pair sampling on the device: host side of csrc/pair_sample.hip (disgat_pair_sample_plan / _emit).

The reference builds dense N x N masks (pretrainer.py:683-707, 524-576):
    mask = (rand(N,N) < 3*rho) | first third of the shuffled positives;  indices = mask.nonzero().T
with rho = n_pos / N^2.  The kernels draw the same distribution, entry for entry, in O(output) work: every row's column
range is a Bernoulli(3 rho) process walked by geometric skipping (one wave per column interval, 64 sorted columns per
round), the "third of the positives" is an exact-size uniform subset picked by a keyed pseudo-random permutation, the two
sorted sets are merged in the wave and the labels (membership in the positive set) fall out of the merge.  The
generator's (seed, step) live on the device: a train_step captured in a HIP graph draws a fresh list on every replay
with no host input at all.  This module keeps what the kernels need per positive set - its CSR form, the work-item
table, scratch - and offers the list in two forms: exact length (one size read back) and fixed capacity with padding.

On a row shard (parallel.DistGraph) a rank samples over ITS rows only: rows are local, columns global (n_cols = the global
node count) and rho stays the global density, so the union over ranks has the unsharded distribution; every rank selects
a third of its own positives.

There is no CPU path: a sampler over CPU tensors raises (the CPU restatement lives in oracle/sampler_oracle.py, for tests).
"""
import math

import torch

from . import _lib

PCAP, RCAP, RMEAN = 256, 256, 96          # include/disgat_hip.h: DISGAT_SAMPLE_PCAP / _RCAP / _RMEAN
_I64_MAX = (1 << 63) - 1
# The trainers' eager train_step()s (pretrainer.*.sample_train) take their lists at FIXED capacity (sample_padded: the
# valid length stays on the device, the tail is padding the loss kernels skip) instead of exact length: every buffer sized by
# the list - scores, sign records (17 GB at 1M nodes), gradients - then has the same size every step and comes back from the
# caching allocator.  (The eager step's BACKWARD still reads two numbers back while it builds its segment tables -
# ops_bwd._segments: sortedness, item count; only captured steps, whose lists carry `_disgat_static`, take the read-free
# ops_bwd._segments_static - and the padding tail, 8 sigma + overlap, is scored and segmented like real pairs under the last
# key.)  With exact lengths the sizes wander by a few MB from step to step
# and a step occasionally pays for fresh device allocations of tens of GB (a bench.py run read 1 392 ms for a 1 024 ms step).
PADDED_LISTS = True


def flat_edges(graph):
    """Sorted int64 row*n_cols+col of every CSR entry (row local, col global on a shard)."""
    return graph.row * graph.n_cols + graph.col.to(torch.int64)


def membership(flat, pos_flat):
    """float32 0/1: flat[i] in pos_flat (pos_flat sorted)."""
    if pos_flat.numel() == 0:
        return torch.zeros(flat.shape, dtype=torch.float32, device=flat.device)
    j = torch.searchsorted(pos_flat, flat).clamp_(max=pos_flat.numel() - 1)
    return (pos_flat[j] == flat).to(torch.float32)


def build_items(pos_flat, n_rows, n_cols, p):
    """Work items of the sampler kernels, int32 [n_items, 8] = {row, col_lo, col_hi, pos_lo, pos_hi, 0, 0, 0} ordered by
    (row, col_lo): every row's column range cut at the multiples of floor(RMEAN / p) columns (expected random entries per
    item <= RMEAN) and at every PCAP-th positive of the row (positives per item <= PCAP).  Disjoint intervals of a Bernoulli
    process are independent, so any such partition leaves the distribution untouched.  Built once per positive set."""
    dev = pos_flat.device
    npos = int(pos_flat.shape[0])
    lblk = n_cols if p <= 0 else min(n_cols, max(1, int(RMEAN / p)))
    nblk = -(-n_cols // lblk)
    rows = torch.div(pos_flat, n_cols, rounding_mode="floor")
    first = torch.searchsorted(pos_flat, torch.arange(n_rows, device=dev, dtype=torch.int64) * n_cols)     # rowptr[:-1]
    within = torch.arange(npos, device=dev, dtype=torch.int64) - first[rows]
    cut = (within % PCAP == 0) & (within > 0)
    grid = (torch.arange(n_rows, device=dev, dtype=torch.int64)[:, None] * n_cols
            + torch.arange(nblk, device=dev, dtype=torch.int64)[None, :] * lblk).reshape(-1)
    starts = torch.unique(torch.cat([grid, pos_flat[cut]]))
    it_row = torch.div(starts, n_cols, rounding_mode="floor")
    clo = starts - it_row * n_cols
    nxt = torch.cat([starts[1:], starts.new_tensor([n_rows * n_cols])])
    nxt_row = torch.div(nxt, n_cols, rounding_mode="floor")
    chi = torch.where(nxt_row == it_row, nxt - it_row * n_cols, torch.full_like(clo, n_cols))
    plo = torch.searchsorted(pos_flat, starts)
    phi = torch.cat([plo[1:], plo.new_tensor([npos])])
    zero = torch.zeros_like(clo)
    return torch.stack([it_row, clo, chi, plo, phi, zero, zero, zero], 1).to(torch.int32).contiguous()


class PairSampler:
    """The pair lists of one positive set.  pos_flat: sorted unique int64 device tensor row * n_cols + col of this process's
    positives; n_rows: rows this process owns; n_cols: number of nodes a column can name (default n_rows = unsharded);
    n_pos_global: positives over all ranks (default: len(pos_flat)).

    sample()         -> (indices int64 [2, M], labels f32 [M]): rows local, columns global, row-major sorted; reads M back.
    sample_static()  -> the same list in tensors of FIXED capacity (mean + 8 sigma): the tail is padding - pair
                        (n_rows-1, n_cols-1), so the list stays row-major sorted, with label -1, which the pair-loss kernels
                        skip (include/disgat_hip.h: disgat_pair_loss); the valid length stays on the device
                        (labels._disgat_count).  Nothing is read back: capturable in a HIP graph.
    Event counters (device, never reset): `overflow` = items that hit a kernel capacity (never in practice: the item table
    bounds them), `clamped` = lists longer than the fixed capacity (8 sigma).  events() reads both."""

    def __init__(self, n_rows, pos_flat, n_cols=None, n_pos_global=None, seed=None):
        if not pos_flat.is_cuda:
            raise RuntimeError("PairSampler: the pair sampler is a HIP kernel (csrc/pair_sample.hip); there is no CPU path")
        self.n_rows, self.n_cols = int(n_rows), int(n_rows if n_cols is None else n_cols)
        if self.n_rows < 1 or self.n_cols > (1 << 30):
            raise ValueError("PairSampler: {} x {} outside the kernel envelope".format(self.n_rows, self.n_cols))
        dev = pos_flat.device
        self.device = dev
        self.pos = pos_flat
        self.npos = int(pos_flat.shape[0])
        n_glob = self.npos if n_pos_global is None else int(n_pos_global)
        self.p = min(1.0, 3.0 * n_glob / (float(self.n_cols) * float(self.n_cols)))       # pretrainer.py:691-692: edge_ratio * 3
        self.n_sel = self.npos // 3                                                        # pretrainer.py:697-700
        rows = torch.div(pos_flat, self.n_cols, rounding_mode="floor")
        self.pos_col = (pos_flat - rows * self.n_cols).to(torch.int32).contiguous()
        self.items = build_items(pos_flat, self.n_rows, self.n_cols, self.p)
        self.n_items = int(self.items.shape[0])
        self.ipw = max(1, min(16, self.n_items // 8192))          # consecutive items per wave: >= ~2k blocks when there is work for them
        self.n_blocks = -(-self.n_items // (4 * self.ipw))
        self.item_count = torch.empty(self.n_items, dtype=torch.int32, device=dev)
        self.block_count = torch.empty(self.n_blocks, dtype=torch.int32, device=dev)
        self.block_off = torch.empty(self.n_blocks, dtype=torch.int64, device=dev)
        self.meta = torch.zeros(8, dtype=torch.int64, device=dev)
        self._seeded = False
        if seed is not None:
            self.reseed(seed)
        total = float(self.n_rows) * float(self.n_cols)
        mean = total * self.p
        k_max = min(total, math.ceil(mean + 8.0 * math.sqrt(max(mean * (1.0 - self.p), 1.0)) + 8))
        self.capacity = int(k_max) + self.n_sel

    # ---- generator state
    def reseed(self, seed, step=0):
        self.meta[0] = int(seed) & _I64_MAX
        self.meta[1] = int(step)
        self._seeded = True

    def _ensure_seed(self):
        if not self._seeded:      # first use: from torch's CPU generator, so torch.manual_seed / main.reseed_rank govern the stream
            self.reseed(int(torch.randint(0, _I64_MAX, (1,), dtype=torch.int64).item()))

    def events(self):
        """(overflow, clamped) so far - one read-back."""
        m = self.meta[4:6].tolist()
        return int(m[0]), int(m[1])

    # ---- launches
    def _plan(self, capacity, count):
        from .ops import _stream
        _lib.call("disgat_pair_sample_plan", self.items.data_ptr(), self.n_items, self.ipw, self.pos_col.data_ptr(), self.npos,
                  self.n_sel, self.p, int(capacity), self.meta.data_ptr(), self.item_count.data_ptr(), self.block_count.data_ptr(),
                  self.block_off.data_ptr(), None if count is None else count.data_ptr(), _stream())

    def _emit(self, capacity, idx, lab, pad):
        from .ops import _stream
        _lib.call("disgat_pair_sample_emit", self.items.data_ptr(), self.n_items, self.ipw, self.pos_col.data_ptr(), self.npos,
                  self.n_sel, self.p, self.n_rows, self.n_cols, int(capacity), self.meta.data_ptr(), self.item_count.data_ptr(),
                  self.block_off.data_ptr(), idx.data_ptr(), lab.data_ptr(), int(pad), _stream())

    def sample(self):
        self._ensure_seed()
        self._plan(_I64_MAX, None)
        m = int(self.meta[3])                                   # the one host read of a step: the list's length
        idx = torch.empty((2, m), dtype=torch.int64, device=self.device)
        lab = torch.empty(m, dtype=torch.float32, device=self.device)
        self._emit(m, idx, lab, 0)
        idx._disgat_checked = (self.n_rows, self.n_cols, idx._version)    # in range by construction (ops.check_pairs): no round trip
        return idx, lab

    def sample_padded(self):
        """sample_static() for steps that are NOT replayed from a graph: the same fixed-capacity list, without the mark that
        sends the backward to the fixed-capacity segment tables (an eager step may read sizes back and sort its work items)."""
        idx, lab = self.sample_static()
        del idx._disgat_static
        return idx, lab

    def sample_static(self):
        self._ensure_seed()
        cap = self.capacity
        count = torch.empty((), dtype=torch.float64, device=self.device)
        idx = torch.empty((2, cap), dtype=torch.int64, device=self.device)
        lab = torch.empty(cap, dtype=torch.float32, device=self.device)
        self._plan(cap, count)
        self._emit(cap, idx, lab, 1)
        idx._disgat_checked = (self.n_rows, self.n_cols, idx._version)
        idx._disgat_static = True          # ops_bwd: segment structures without host round trips
        lab._disgat_count = count          # pretrainer._pair_loss_value: the mean runs over the valid entries
        return idx, lab


def sample_pairs(n_rows, pos_flat, n_cols=None, n_pos_global=None, seed=None):
    """One list from a throw-away sampler (tools, tests; trainers keep their PairSampler with the graph)."""
    return PairSampler(n_rows, pos_flat, n_cols=n_cols, n_pos_global=n_pos_global, seed=seed).sample()


class FixedList:
    """A given pair list behind the static sampler's interface (tests and callers that bring their own lists to a captured step)."""

    def __init__(self, indices, labels, n_rows=None, n_cols=None):
        self.indices, self.labels = indices.contiguous(), labels.contiguous()
        if n_rows is not None:
            self.indices._disgat_checked = (int(n_rows), int(n_rows if n_cols is None else n_cols), self.indices._version)
        self.indices._disgat_static = True
        self.labels._disgat_count = torch.full((), float(labels.shape[0]), dtype=torch.float64, device=labels.device)

    def sample_static(self):
        return self.indices, self.labels

    sample = sample_static

    def events(self):
        return 0, 0
