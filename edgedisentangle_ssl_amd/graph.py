"""Device-resident CSR view of the adjacency's index set + wave work items.

The reference re-derives the edge list with ``adj.coalesce().indices()`` inside
every head of every layer (layers.py:344; 16 sorts per forward).  DISGAT uses
only the INDEX SET of the adjacency (its values are ignored), so the graph is
preprocessed once per adjacency tensor into

  rowptr int32 [N+1], col int32 [E]   CSR; CSR edge order == coalesced row-major
                                      order, so per-edge outputs need no permutation
  items  int32 [n_items, 4]           {row, e_begin, e_end, slot}: one wave64 each.
                                      Rows longer than ``chunk`` edges are cut into
                                      near-equal slices (slot >= 0 -> partial record,
                                      summed by disgat_edge_combine); empty rows keep one
                                      empty item so their output row is written as zeros.
                                      Items are ordered by descending length so the hub
                                      slices start first (power-law tail).
  t_ptr/t_src/t_eid                   the transposed structure (CSC) with the forward edge
                                      id of every entry - gather-only backward passes.
"""
import weakref

import torch


class WorkItems:
    __slots__ = ("items", "n_items", "split_rows", "split_ptr", "n_split", "n_slots", "chunk", "totals")


def build_items(ptr, chunk):
    """Wave work items over the segments [ptr[i], ptr[i+1]): one item per segment, segments longer
    than `chunk` cut into near-equal slices (slot >= 0), empty segments keep one empty item (their
    output row must still be written), items ordered by descending length."""
    dev = ptr.device
    rp = ptr.to(torch.int64)
    n = rp.numel() - 1
    deg = rp[1:] - rp[:-1]
    nchunk = torch.clamp((deg + chunk - 1) // chunk, min=1)
    n_items = int(nchunk.sum())
    item_row = torch.repeat_interleave(torch.arange(n, device=dev), nchunk)
    first = torch.cumsum(nchunk, 0) - nchunk
    j = torch.arange(n_items, device=dev) - first[item_row]
    size = (deg // nchunk)[item_row]
    rem = (deg % nchunk)[item_row]
    begin = rp[:-1][item_row] + j * size + torch.minimum(j, rem)
    end = begin + size + (j < rem).to(torch.int64)
    is_split = (nchunk > 1)[item_row]
    slot = torch.where(is_split, torch.cumsum(is_split.to(torch.int64), 0) - 1, torch.full_like(j, -1))
    items = torch.stack([item_row, begin, end, slot], 1)
    order = torch.sort(end - begin, descending=True, stable=True).indices
    wi = WorkItems()
    wi.items = items[order].to(torch.int32).contiguous()
    wi.n_items = n_items
    split_rows = torch.nonzero(nchunk > 1)[:, 0]
    wi.n_split = int(split_rows.shape[0])
    wi.split_rows = split_rows.to(torch.int32).contiguous()
    sp = torch.zeros(wi.n_split + 1, dtype=torch.int64, device=dev)
    if wi.n_split:
        sp[1:] = torch.cumsum(nchunk[split_rows], 0)
    wi.split_ptr = sp.to(torch.int32).contiguous()
    wi.n_slots = int(sp[-1])
    wi.chunk = chunk
    return wi


class CSRGraph:
    def __init__(self, n, rowptr, col, row):
        self.n = int(n)
        self.rowptr = rowptr
        self.col = col
        self.row = row                      # int64 [E] (kept for host-side consumers / tests)
        self.n_cols = self.n                # number of nodes a column id can name (> n on a row shard)
        self.row_start = 0                  # global id of local row 0 (> 0 on a row shard)
        self.nnz = int(col.shape[0])
        self.device = col.device
        self._items = {}
        self._transpose = None
        self._pairs = None

    # ------------------------------------------------------------------ construction
    @staticmethod
    def from_index(indices, n):
        """indices: int64 [2, nnz_raw] (possibly unsorted, with duplicates)."""
        if indices.numel():
            lo, hi = torch.aminmax(indices)          # once per adjacency: a bad id would be a GPU fault later
            if int(lo) < 0 or int(hi) >= n:
                raise ValueError(f"adjacency index out of range: ids in [{int(lo)},{int(hi)}] for {n} nodes")
        flat = torch.unique(indices[0].to(torch.int64) * n + indices[1].to(torch.int64))
        row = torch.div(flat, n, rounding_mode="floor")
        col = (flat - row * n).to(torch.int32)
        counts = torch.bincount(row, minlength=n)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=flat.device)
        rowptr[1:] = torch.cumsum(counts, 0)
        if int(rowptr[-1]) >= 2 ** 31:
            raise ValueError("more than 2^31 edges in one partition")
        return CSRGraph(n, rowptr.to(torch.int32).contiguous(), col.contiguous(), row.contiguous())

    @staticmethod
    def from_adj(adj):
        """adj: torch sparse COO [N,N] (may be uncoalesced), as produced by the
        reference's load_data (data_load.py:158-165)."""
        if not adj.is_sparse:
            raise NotImplementedError("dense adjacency (no --sparse) is outside the HIP path (SURVEY 2 #16)")
        n = adj.shape[0]
        idx = adj.indices() if adj.is_coalesced() else adj._indices()
        return CSRGraph.from_index(idx, n)

    def indices(self):
        """(2,E) int64 == adj.coalesce().indices()."""
        return torch.stack([self.row, self.col.to(torch.int64)])

    def edge_pairs(self):
        """The edge list as a pair list for the aux scorer (built once; in range by construction)."""
        if self._pairs is None:
            self._pairs = self.indices().contiguous()
            self._pairs._disgat_checked = (int(self.n), int(getattr(self, "n_cols", self.n)), self._pairs._version)
            self._pairs._disgat_static = True      # backward segment tables without host round trips (ops_bwd._segments_static):
                                                   # the list is a fixed part of the graph, also inside captured steps
        return self._pairs

    # ------------------------------------------------------------------ work items
    def work_items(self, chunk):
        wi = self._items.get(chunk)
        if wi is None:
            wi = build_items(self.rowptr, chunk)
            self._items[chunk] = wi
        return wi

    # ------------------------------------------------------------------ transpose (for backward)
    def transpose(self):
        """CSC of the same index set: t_ptr int32 [N+1]; for every column c the
        entries t_src (row of the forward edge) and t_eid (forward edge id)."""
        if self._transpose is None:
            colL = self.col.to(torch.int64)
            order = torch.sort(colL, stable=True).indices
            counts = torch.bincount(colL, minlength=self.n_cols)
            tp = torch.zeros(self.n_cols + 1, dtype=torch.int64, device=self.device)
            tp[1:] = torch.cumsum(counts, 0)
            t = CSRGraph(self.n_cols, tp.to(torch.int32).contiguous(), self.row[order].to(torch.int32).contiguous(),
                         colL[order].contiguous())
            t.eid = order.to(torch.int32).contiguous()
            self._transpose = t
        return self._transpose


_GRAPH_CACHE = {}      # id(adj tensor object) -> (weakref to that object, CSRGraph)


def graph_of(adj):
    """CSRGraph for an adjacency tensor, built once per tensor OBJECT (the reference hands the same
    `adj` to every call, main.py:272-345).  Entries hold a weak reference and are validated by
    identity, so a freed tensor whose storage address gets recycled for another graph can never
    alias a stale CSR; they disappear when the tensor is collected."""
    if isinstance(adj, CSRGraph):
        return adj
    key = id(adj)
    hit = _GRAPH_CACHE.get(key)
    if hit is not None and hit[0]() is adj:
        return hit[1]
    g = CSRGraph.from_adj(adj)
    try:
        ref = weakref.ref(adj, lambda _r, k=key: _GRAPH_CACHE.pop(k, None))
    except TypeError:          # not weak-referenceable: do not cache
        return g
    _GRAPH_CACHE[key] = (ref, g)
    return g
