"""Row-range sharding of the DISGAT path across the GPUs of one MI355X node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  Softmax and aggregation are row-local, so rank g owns a contiguous range of CSR rows
(balanced by nnz, not by row count: power-law skew) and needs, per layer, only the COLUMN-side
operands of the neighbours its rows reference.  For the random-column graphs of BASELINE.json the
halo is ~all of N, so the exchange is one all-gather of the layer input x per layer
(N*F_in*4 bytes in total, 1/G of it sent by each rank - not an all-reduce of zero-padded buffers,
which would move G x more); the column-side score operand (att 3: Q = x W_bot, H*F_out wide) is
then recomputed locally from the gathered x instead of being exchanged (8x fewer bytes on the
per-link-bound xGMI mesh; the redundant GEMM is cheaper than 7 links x ~50 GB/s).  Loss terms
reduce with one all-reduce of a few doubles; in training the replicated parameters' gradients are
summed with one bucketed all-reduce per step (classic data parallelism).
"""
import torch
import torch.distributed as dist

from .graph import CSRGraph


def balanced_row_ranges(rowptr, world):
    """Row boundaries [world+1] such that every range holds ~nnz/world entries."""
    rp = rowptr.to(torch.int64)
    n = rp.numel() - 1
    nnz = int(rp[-1])
    targets = torch.arange(1, world, device=rp.device, dtype=torch.int64) * nnz // world
    cuts = torch.searchsorted(rp, targets, right=False).clamp_(0, n)
    b = torch.cat([torch.zeros(1, dtype=torch.int64, device=rp.device), cuts,
                   torch.full((1,), n, dtype=torch.int64, device=rp.device)])
    return torch.cummax(b, 0).values


class DistGraph(CSRGraph):
    """CSR of the rows this rank owns: LOCAL row ids, GLOBAL column ids."""

    def __init__(self, n_local, rowptr, col, row, n_global, row_start, counts, group=None):
        super().__init__(n_local, rowptr, col, row)
        self.n_global = int(n_global)
        self.n_cols = self.n_global
        self.row_start = int(row_start)
        self.counts = [int(c) for c in counts]          # rows owned by every rank
        self.group = group
        self.world = len(self.counts)

    @staticmethod
    def from_local_edges(local_rows, global_cols, n_local, n_global, row_start, counts, group=None):
        flat = torch.unique(local_rows.to(torch.int64) * n_global + global_cols.to(torch.int64))
        row = torch.div(flat, n_global, rounding_mode="floor")
        col = (flat - row * n_global).to(torch.int32)
        rp = torch.zeros(n_local + 1, dtype=torch.int64, device=flat.device)
        rp[1:] = torch.cumsum(torch.bincount(row, minlength=n_local), 0)
        return DistGraph(n_local, rp.to(torch.int32).contiguous(), col.contiguous(), row.contiguous(), n_global,
                         row_start, counts, group)

    @staticmethod
    def shard(graph, rank, world, group=None):
        """Cut a replicated global CSRGraph into this rank's row range (tests / small graphs)."""
        b = balanced_row_ranges(graph.rowptr, world)
        lo, hi = int(b[rank]), int(b[rank + 1])
        rp = graph.rowptr.to(torch.int64)
        e0, e1 = int(rp[lo]), int(rp[hi])
        counts = (b[1:] - b[:-1]).tolist()
        return DistGraph(hi - lo, (rp[lo:hi + 1] - e0).to(torch.int32).contiguous(), graph.col[e0:e1].contiguous(),
                         (graph.row[e0:e1] - lo).contiguous(), graph.n, lo, counts, group)


# ---------------------------------------------------------------------------------------------------------------
# Pipelined all-gather ("overlap with the row-side GEMM").  The layer input is gathered in EXCHANGE_SLICES
# asynchronous collectives - slice j carries rows [j/S, (j+1)/S) of EVERY rank's range, so each collective uses all
# links of the xGMI mesh like the one big all-gather did (a per-peer split would light one source's links at a time) -
# and the consumers start on what has arrived: the rank's own rows need no wait at all (P = x_local W_top and the
# own-row block of Q run while slice 0 is on the links), then Q[rows of slice j] = x_all[rows of slice j] W_bot as each
# slice lands (layers._pack_score_operands -> project_gathered).  With the nccl (= RCCL) backend an async collective
# runs on the process group's own stream and work.wait() only makes the compute stream wait for it: the host never
# blocks.  Ragged (nnz-balanced) ranges: every slice is padded to its longest member - one padded
# all_gather_into_tensor per slice, no tensor lists, no torch.cat.
def exchange_slices():
    import os
    return max(1, int(os.environ.get("DISGAT_EXCHANGE_SLICES", "4")))


class PendingGather:
    """The slices of one gathered table still on the links.  It hangs on the table itself (attribute `_disgat_pending`,
    set by _gather_rows) - no registry keyed by an address that a later tensor could inherit; it holds the table's storage
    through a raw alias only, so table and gather form no reference cycle and an abandoned gather (an exception between
    its start and wait_all) dies with the table."""

    def __init__(self, x_local, counts, group, n_slices):
        world, rank = len(counts), dist.get_rank(group)
        self.counts, self.rank, self.group = counts, rank, group
        self.offsets = [0]
        for c in counts:
            self.offsets.append(self.offsets[-1] + c)
        feat = tuple(x_local.shape[1:])
        self.x_all = x_local.new_empty((self.offsets[-1],) + feat)
        self.x_all[self.offsets[rank]: self.offsets[rank + 1]] = x_local              # own rows: valid at once
        self.own = (self.offsets[rank], self.offsets[rank + 1])
        self.slices = []
        S = max(1, min(n_slices, max(counts) if counts else 1))
        for j in range(S):
            lo = [(c * j) // S for c in counts]
            sz = [(c * (j + 1)) // S - l for c, l in zip(counts, lo)]
            ms = max(sz)
            if ms == 0:
                continue
            inp = x_local[lo[rank]: lo[rank] + sz[rank]]
            if sz[rank] != ms:                                   # ragged: pad this rank's slice to the longest
                pad = x_local.new_zeros((ms,) + feat)
                pad[: sz[rank]] = inp
                inp = pad
            tmp = x_local.new_empty((world * ms,) + feat)
            work = dist.all_gather_into_tensor(tmp, inp.contiguous(), group=group, async_op=True)
            self.slices.append((work, tmp, ms, lo, sz))
        self.next = 0
        self.raw = self.x_all            # fills go through this alias (replaced by x_all.data once autograd owns x_all)

    @property
    def done(self):
        return self.next >= len(self.slices)

    def release(self):
        """Called once _AllGatherRows.forward has returned x_all: keep the storage, not the autograd output."""
        self.raw = self.x_all.data
        self.x_all = None

    def wait_next(self):
        """Wait (stream-level) for the next slice and copy the peers' rows into place.  Returns the global row ranges
        [(lo, hi), ...] that became valid, or None when nothing is pending any more."""
        if self.next >= len(self.slices):
            return None
        work, tmp, ms, lo, sz = self.slices[self.next]
        self.slices[self.next] = None
        self.next += 1
        work.wait()
        ranges = []
        for r, (l, n) in enumerate(zip(lo, sz)):
            if r == self.rank or n == 0:
                continue
            g0 = self.offsets[r] + l
            # a raw fill of the buffer, never an autograd op: by now x_all is the OUTPUT of _AllGatherRows, and a tracked
            # in-place copy would cut these rows out of its backward (their gradient must reach the reduce-scatter)
            self.raw[g0: g0 + n] = tmp[r * ms: r * ms + n]
            ranges.append((g0, g0 + n))
        return ranges

    def wait_all(self):
        while self.wait_next() is not None:
            pass


def pending_of(x_all):
    """The PendingGather still filling `x_all`, or None (unsharded / complete / halo table)."""
    pend = getattr(x_all, "_disgat_pending", None) if torch.is_tensor(x_all) else None
    return pend if (pend is not None and not pend.done) else None


def finish(x_all):
    """Make every row of a gathered table valid for the kernels enqueued after this call."""
    pend = pending_of(x_all)
    if pend is not None:
        pend.wait_all()
    return x_all


class GatherAdjoint:
    """Adjoint of the all-gather, in flight: every rank holds a gradient for ALL rows; the owner of a row range needs their
    sum over ranks = one reduce-scatter (each rank receives 1/G of what an all-reduce would move), ragged ranges padded to
    the longest.  Issued asynchronously; result() makes the compute stream (not the host) wait for it."""

    def __init__(self, g, counts, group):
        rank, world = dist.get_rank(group), len(counts)
        self.src_version = g._version
        g = g.contiguous()
        if len(set(counts)) == 1:
            self.out = g.new_empty((counts[0],) + tuple(g.shape[1:]))
            self.keep = g
            self.work = dist.reduce_scatter_tensor(self.out, g, group=group, async_op=True)
            return
        mx = max(counts)                          # ragged (nnz-balanced) ranges: pad every range to the longest
        pad = g.new_zeros((world, mx) + tuple(g.shape[1:]))
        off = 0
        for r, c in enumerate(counts):
            pad[r, :c] = g[off: off + c]
            off += c
        out = g.new_empty((mx,) + tuple(g.shape[1:]))
        self.keep = pad
        self.work = dist.reduce_scatter_tensor(out, pad.view((world * mx,) + tuple(g.shape[1:])), group=group, async_op=True)
        self.out = out[: counts[rank]]

    def result(self):
        self.work.wait()
        self.keep = None
        return self.out


ADJOINT_EARLY_STARTS = 0        # how many adjoints were put on the links by their producer (tests read this)
ADJOINT_EARLY_TAKEN = 0         # ... and how many of those _AllGatherRows.backward picked up (tests: started == taken)
_ADJOINT_OPEN = []              # early adjoints nobody has collected yet: an orphan is waited for before the plain path runs


def start_adjoint(x_all, g):
    """Called by the backward that has just finished g = d loss / d x_all (ops_bwd.layer_backward_u: the data-gradient GEMM
    of the column-side score operand, plus the aggregation's share) BEFORE it goes on with work that does not touch g (its
    weight-gradient GEMM): the reduce-scatter starts now and runs on the links under that work; _AllGatherRows.backward
    picks the result up.  No-op for tensors that are not a gathered table."""
    global ADJOINT_EARLY_STARTS
    info = getattr(x_all, "_disgat_gather", None) if torch.is_tensor(x_all) else None
    if info is None or g is None or tuple(g.shape) != tuple(x_all.shape):
        return False
    g._disgat_adjoint = GatherAdjoint(g, *info)
    _ADJOINT_OPEN.append(g._disgat_adjoint)
    ADJOINT_EARLY_STARTS += 1
    return True


class _AllGatherRows(torch.autograd.Function):
    _handoff = []           # forward -> _gather_rows: the PendingGather of the table just returned

    @staticmethod
    def forward(ctx, x, counts, group, n_slices=1, leave_pending=False):
        ctx.counts, ctx.group = counts, group
        pend = PendingGather(x, counts, group, n_slices)
        if not leave_pending:
            pend.wait_all()
        _AllGatherRows._handoff.append(pend)
        return pend.x_all

    @staticmethod
    def backward(ctx, g):
        global ADJOINT_EARLY_TAKEN
        early = getattr(g, "_disgat_adjoint", None)
        if early is not None and early.src_version == g._version:      # already on the links (start_adjoint)
            g._disgat_adjoint = None
            if early in _ADJOINT_OPEN:
                _ADJOINT_OPEN.remove(early)
            ADJOINT_EARLY_TAKEN += 1
            return early.result(), None, None, None, None
        # the gradient is not the tensor the producer tagged (autograd summed several contributions or re-wrapped it): an early
        # reduce-scatter may be in flight for a PART of it - wait for it (its result is dropped) before the whole sum goes out
        while _ADJOINT_OPEN:
            _ADJOINT_OPEN.pop().result()
        return GatherAdjoint(g, ctx.counts, ctx.group).result(), None, None, None, None


def _gather_rows(x_local, counts, group, n_slices, leave_pending):
    """_AllGatherRows.apply + what hangs on its output: the gather still pending (pending_of) and the marker that lets
    the consumer's backward start the adjoint early (start_adjoint)."""
    del _AllGatherRows._handoff[:]
    x_all = _AllGatherRows.apply(x_local, counts, group, n_slices, leave_pending)
    pend = _AllGatherRows._handoff.pop()
    pend.release()
    x_all._disgat_pending = None if pend.done else pend
    x_all._disgat_gather = (counts, group)
    return x_all


class _ProjectGathered(torch.autograd.Function):
    """q = x_all @ w for a gathered table that may still be arriving: the rank's own rows first, then every slice's
    rows as it lands - the redundant column-side GEMM of the sharded path runs under the links' time instead of
    after it.  Backward = the plain linear backward (the all-gather's adjoint then reduce-scatters grad x_all)."""

    @staticmethod
    def forward(ctx, x_all, w, a_amax, w_split):
        from . import ops_gemm
        pend = pending_of(x_all)
        q = x_all.new_empty((x_all.shape[0], w.shape[1]))

        def block(lo, hi):
            if hi > lo:
                ops_gemm._forward(x_all[lo:hi], w, None, None, ops_gemm.ACT_NONE, 0.0, a_amax, w_split, out=q[lo:hi])

        if pend is None:
            block(0, x_all.shape[0])
        else:
            block(*pend.own)
            while True:
                ranges = pend.wait_next()
                if ranges is None:
                    break
                for lo, hi in ranges:
                    block(lo, hi)
        ctx.save_for_backward(x_all, w)
        ctx.a_amax = a_amax
        return q

    @staticmethod
    def backward(ctx, g):
        from . import ops_gemm
        x_all, w = ctx.saved_tensors
        ga, gw = ops_gemm.linear_backward(x_all, w, g.contiguous(), ctx.a_amax, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return ga, gw, None, None


def project_gathered(x_all, w, a_amax=None, w_split=None):
    return _ProjectGathered.apply(x_all, w, a_amax, w_split)


def all_gather_rows(x_local, graph):
    """[n_local,F] on every rank -> [n_global,F] (rank order = row order).  Identity when unsharded."""
    if not isinstance(graph, DistGraph) or graph.world == 1:
        return x_local
    return _gather_rows(x_local, graph.counts, graph.group, exchange_slices(), False)


# ---------------------------------------------------------------------------------------------------------------
# Halo-only exchange (north_star: "halo features"): a rank fetches just the rows its edges reference instead of all
# N_global rows.  Worth it when the referenced set is a small part of the graph (graphs with locality / many ranks);
# on BASELINE.json's random-column generator a rank references 71-92 % of all nodes, and a pass that scores auxiliary
# pairs (uniform random columns) needs every row anyway, so `exchange()` picks per call:
#   DISGAT_EXCHANGE=auto (default): halo for edge-only passes when the referenced set is < 60 % of the graph,
#   =halo: halo for every edge-only pass, =allgather: never.
def _a2a(out, inp, out_splits, in_splits, group):
    """all_to_all_single; gloo moves CUDA tensors through the host (the rehearsal / test configuration)."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=group)
    return out


class HaloPlan:
    """Built once per DistGraph: which global rows this rank's edges reference (`ref`, sorted = ordered by owner),
    which of its own rows every peer wants (`send_idx`, `send_counts`), how many rows arrive from every peer
    (`recv_counts`), and the graph re-indexed to the compact table (`graph_c`: column ids = positions in `ref`)."""

    def __init__(self, graph):
        dev = graph.col.device
        world, group = graph.world, graph.group
        ref = torch.unique(graph.col.to(torch.int64))
        bounds = torch.cat([torch.zeros(1, dtype=torch.int64), torch.tensor(graph.counts, dtype=torch.int64).cumsum(0)]).to(dev)
        owner = torch.searchsorted(bounds, ref, right=True) - 1
        recv_counts = torch.bincount(owner, minlength=world)
        send_counts = torch.empty_like(recv_counts)
        _a2a(send_counts, recv_counts, [1] * world, [1] * world, group)
        self.recv_counts = recv_counts.tolist()
        self.send_counts = send_counts.tolist()
        want = torch.empty(sum(self.send_counts), dtype=torch.int64, device=dev)      # ids the peers want from me
        _a2a(want, ref, self.send_counts, self.recv_counts, group)
        self.send_idx = (want - graph.row_start).contiguous()
        assert self.send_idx.numel() == 0 or (int(self.send_idx.min()) >= 0 and int(self.send_idx.max()) < graph.n)
        # adjoint: the rows every peer sends back are summed per owner row in a FIXED order (sorted once here, segmented
        # sum in the backward) - index_add_'s float atomics made the halo path's gradients vary run to run
        self.back_order = torch.sort(self.send_idx, stable=True).indices
        self.back_lengths = torch.bincount(self.send_idx, minlength=graph.n)
        self.ref = ref
        self.n_ref = int(ref.numel())
        self.group = group
        col_c = torch.searchsorted(ref, graph.col.to(torch.int64)).to(torch.int32)
        gc = CSRGraph(graph.n, graph.rowptr, col_c.contiguous(), graph.row)
        gc.n_cols = self.n_ref
        gc.row_start = graph.row_start
        self.graph_c = gc


class _HaloRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan):
        ctx.plan = plan
        out = x.new_empty((plan.n_ref,) + tuple(x.shape[1:]))
        _a2a(out, x[plan.send_idx], plan.recv_counts, plan.send_counts, plan.group)
        return out

    @staticmethod
    def backward(ctx, g):
        """Adjoint: every referenced row's gradient travels back to its owner, which sums what it receives."""
        plan = ctx.plan
        back = g.new_empty((plan.send_idx.numel(),) + tuple(g.shape[1:]))
        _a2a(back, g.contiguous(), plan.send_counts, plan.recv_counts, plan.group)
        if back.shape[0] == 0:
            return g.new_zeros((plan.graph_c.n,) + tuple(g.shape[1:])), None
        gx = torch.segment_reduce(back[plan.back_order], "sum", lengths=plan.back_lengths, axis=0, initial=0.0)
        return gx, None


# Static inputs.  The node features a run trains on never change, yet every encoder pass would all-gather them again
# (3 of the 6 exchanges of a T_iter step; every layer-1 exchange of a training run without input dropout).  A caller
# can declare a tensor static (`mark_static(features)`, as main.run and bench.py do); its gathered form is then kept
# with the graph and reused while the tensor's storage and version counter are unchanged and no gradient is asked of
# it.  This caches a COMMUNICATION result of an unchanged input, never computed outputs.
_STATIC = {}        # storage address -> (weakref to the marked tensor, version at marking)


def mark_static(x):
    import weakref
    _STATIC[x.data_ptr()] = (weakref.ref(x), x._version)
    return x


def _static_key(x):
    ent = _STATIC.get(x.data_ptr())
    if ent is None or x.requires_grad:
        return None
    ref, ver = ent
    base = ref()
    if base is None:                       # the marked tensor is gone: its address may belong to something else now
        del _STATIC[x.data_ptr()]
        return None
    if base._version != ver or x._version != base._version or x.shape != base.shape or x.stride() != base.stride():
        return None
    return (x.data_ptr(), ver, tuple(x.shape))


def exchange(x_local, graph, edge_only, pipelined=False):
    """The per-layer exchange.  Returns (x_cols, graph_eff): the operand table column ids of `graph_eff`
    index.  Unsharded: (x_local, graph).  Sharded: the full all-gather (graph unchanged), or - for passes that score
    no auxiliary pairs - the halo exchange with the compact re-indexed graph.
    pipelined=True: the all-gather is returned while its slices are still on the links (pending_of(x_cols) is then
    not None); the caller consumes it through project_gathered() / finish()."""
    if not isinstance(graph, DistGraph) or graph.world == 1:
        return x_local, graph
    import os
    mode = os.environ.get("DISGAT_EXCHANGE", "auto")
    if edge_only and mode != "allgather":
        plan = graph.__dict__.get("_halo")
        if plan is None:
            plan = graph._halo = HaloPlan(graph)
            frac = torch.tensor([plan.n_ref / max(1, graph.n_global)], dtype=torch.float64, device=x_local.device)
            dist.all_reduce(frac, op=dist.ReduceOp.MAX, group=graph.group)            # one decision for all ranks
            plan.worth_it = float(frac) < 0.6
        if mode == "halo" or plan.worth_it:
            return _HaloRows.apply(x_local, plan), plan.graph_c
    key = _static_key(x_local)
    if key is not None:
        hit = graph.__dict__.get("_static_gather")
        if hit is not None and hit[0] == key:
            return hit[1], graph
        x_all = _gather_rows(x_local, graph.counts, graph.group, exchange_slices(), False)
        graph._static_gather = (key, x_all.detach())
        return x_all, graph
    return _gather_rows(x_local, graph.counts, graph.group, exchange_slices(), bool(pipelined)), graph


_SEEN_FLAGS = {}


def all_reduce_grads(modules, graph):
    """Data-parallel gradient reduction: every rank holds the replicated parameters and the gradient
    contribution of its own rows / pairs; the global gradient is their sum.  The bucket covers EVERY parameter
    that requires grad in module order (a rank whose shard has no edges / pairs gets None gradients from the
    backward nodes; zeros stand in for them), so its layout is identical on all ranks; a has-grad flag per parameter
    rides in the same bucket."""
    if not (isinstance(graph, DistGraph) and graph.world > 1):
        return
    params = [p for m in modules for p in m.parameters() if p.requires_grad]
    if not params:
        return
    local = tuple(p.grad is not None for p in params)
    has = torch.tensor([1.0 if h else 0.0 for h in local], dtype=params[0].dtype, device=params[0].device)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params] + [has])
    dist.all_reduce(flat, group=graph.group)       # one bucket: a few MB of parameters + one has-grad flag each
    # Which parameters got a gradient on SOME rank decides, for the ones with none here, between "stays None" (the
    # optimiser skips it, as in the unsharded run) and "receives the others' sum".  Only those need the flags on the
    # host; which parameters no rank ever touches is structural (the encoder's own unused fusers), so the flags are read
    # back once per (parameter set, local pattern) instead of every step - the step stays free of device syncs.
    if all(local):
        seen = [1.0] * len(params)
    else:
        key = (tuple(id(p) for p in params), local)
        seen = _SEEN_FLAGS.get(key)
        if seen is None:
            seen = _SEEN_FLAGS[key] = flat[-len(params):].tolist()
    off = 0
    for p, any_rank in zip(params, seen):
        g = flat[off: off + p.numel()].view_as(p)
        off += p.numel()
        if p.grad is not None:
            p.grad.copy_(g)
        elif any_rank > 0:                         # unused on every rank (e.g. the encoder's own fusers): stays None,
            p.grad = g.clone()                     # so the optimiser skips it exactly as in the unsharded run


def all_reduce_max(t, graph):
    if isinstance(graph, DistGraph) and graph.world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=graph.group)
    return t


def all_reduce_sum(t, graph):
    if isinstance(graph, DistGraph) and graph.world > 1:
        dist.all_reduce(t, group=graph.group)
    return t
