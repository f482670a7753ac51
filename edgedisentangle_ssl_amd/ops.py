"""torch.autograd.Function wrappers around the C ABI (include/disgat_hip.h).

PyTorch is plumbing here: it owns the device buffers and the stream; the
launchers borrow raw pointers for the duration of a launch.  Every op requires
CUDA (ROCm) tensors and the in-tree libdisgat_hip.so - there is no fallback.
"""
import torch

from . import _lib
from .graph import CSRGraph

# max edges per work item, by attention type: bounds the tail on power-law rows while keeping
# the partial-record traffic (H*F_in floats per chunk) small next to the chunk's gather bytes.  att 3: 512 (round 1: 128;
# same-box T_fwd at C4 with 64 / 128 / 256 / 512 / 1024 / 4096: 96.7 / 95.4 / 95.1 / 94.9 and 102.2 / 101.6 / 101.5 / 101.9 ms
# on a slower box - the hub rows' slices start first, so coarser slices cost no tail and save partial records).
CHUNK = {1: 256, 2: 256, 3: 512, 4: 512}
_DEFAULT_CHUNK = dict(CHUNK)
# Small graphs (Cora, chameleon: 1e4-1e5 edges) have fewer rows than the chip has wave slots and a hub row of a few hundred
# edges is then the whole launch's critical path: below SMALL_ITEMS default-sized items the slices shrink until about that
# many exist, but not below 4 H edges (a slice's partial record is H*F_in floats: keep it well under what the slice gathers).
SMALL_ITEMS = 8192


def chunk_small(att, count, H):
    """Entries per work item for a list of `count` entries: CHUNK[att] (always, when a caller overrode it), smaller for
    small lists."""
    c = CHUNK[att]
    if c != _DEFAULT_CHUNK.get(att) or count >= SMALL_ITEMS * c:
        return c
    small = max(16, 4 * int(H), 1 << max(0, (int(count) // SMALL_ITEMS).bit_length() - 1))
    return min(c, small)


def chunk_for(att, graph, H):
    return chunk_small(att, graph.nnz, H)


# bench.py sets PROFILE to a list: every launch is then bracketed by HIP events recorded on the
# launch stream and logged as (kernel, algorithmic_bytes, start_event, end_event).
PROFILE = None


def _launch(name, label, nbytes, *args):
    if PROFILE is None:
        _lib.call(name, *args)
        return
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    _lib.call(name, *args)
    e.record()
    PROFILE.append((label, nbytes, s, e))


def edge_algorithmic_bytes(att, n, e, H, F_in, F_out):
    """SURVEY 8(d) B_layer: col index + neighbour row + column-side score operand + edge_e store per
    edge; layer input read once + per-head output written once per node; rowptr.  (att 2 gathers no
    column-side operand in this dataflow: S = 0.)"""
    s = {1: H, 2: 0, 3: H * F_out, 4: H * F_out}[att]
    return e * (4 + 4 * F_in + 4 * s + 4 * H) + n * (4 * F_in + 4 * H * F_out) + 4 * (n + 1)


def aux_algorithmic_bytes(att, n, m, nheads, F_in, F_out):
    """SURVEY 8(d) B_aux: index pair + column-side operand per pair; row-side operand once per node."""
    s = {1: nheads, 2: F_in, 3: nheads * F_out, 4: nheads * F_out}[att]
    return m * (8 + 4 * s) + n * 4 * s


# att 3 backward from the forward's sign record (csrc/edge_bwd.hip: seg_grad_sign_kernel) instead of a second
# gather of the operand rows: ~28x fewer bytes per edge in the score backward for 256 B per (edge | pair) of
# extra state between forward and backward.  False selects the gather-only kernels (no extra state).
SIGN_BACKWARD = True


def sign_record(att, H, F_out, count, dev):
    """int32 [count, 64] buffer for the att-3 sign words (layout: include/disgat_hip.h `sign_bits`)."""
    return torch.empty((count, 64), dtype=torch.int32, device=dev)


def wants_sign(att, *operands):
    """True when a differentiable att-3 pass should record signs (decided outside autograd.Function.forward,
    where grad mode is always off)."""
    return (SIGN_BACKWARD and att == 3 and torch.is_grad_enabled()
            and any(t is not None and t.requires_grad for t in operands))


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---- side stream for the pair scorer (no-graph forwards).  Inside one encoder pass the aux scorer of layer l depends only
# on that layer's score operands (P_l, Q_l) and nobody reads its scores before the loss, while the chain the next layer
# waits for - edge pass -> projection + fuser -> P_{l+1}, Q_{l+1} - is half dense GEMMs that leave HBM idle.  The scorer is
# HBM-bound (6.3 TB/s with every wave slot of the chip taken), so launched beside them on an ordinary stream it fills every
# CU and the GEMM workgroups wait for one to drain (round 4's three-stream experiment: -1.5 %).  Here it runs on a stream
# whose CU MASK leaves a share of the chip's CUs to the main stream (hipExtStreamCreateWithCUMask).
# MEASURED AND LEFT OFF (DISGAT_OVERLAP=1 enables it; DISGAT_SIDE_CUS = the scorer's share of the CUs): T_iter 553-578 ms
# sequential against 566 / 667 / 714 / 618 ms at shares 0.75 / 0.625 / 0.875 / 0.5, with single runs at 1.86 s
# (gpurun_out/r05/bench_overlap_2.log).  The kernels do run side by side then - the edge pass takes 1.65x, the back-to-back
# GEMM 3.5x its time alone: a workgroup of it needs a CU's whole register file (512 registers x 4 waves), so it is placed
# only on CUs the scorer's mask leaves EMPTY, i.e. on a quarter of the chip, and what it gains from the idle HBM cycles it
# loses to running on 64 CUs; meanwhile the edge pass of the same chain shares the HBM with the scorer and finishes later.
# The step is at 94 % of its all-traffic floor; no split of the chip between an HBM-bound and a CU-hungry kernel beat the
# sequential order.
_SIDE = {}
_SIDE_PENDING = []


def overlap_enabled():
    """The masked stream is a BLOCKING stream (hipExtStreamCreateWithCUMask takes no flags): it and the legacy default stream
    wait for each other's work, so beside the default stream nothing overlaps and the scorer only loses CUs (measured: 548 ->
    605-628 ms).  The overlap therefore engages only when the caller runs on a stream of its own (torch.cuda.set_stream /
    with torch.cuda.stream(...): bench.py and main.run do)."""
    import os
    return (os.environ.get("DISGAT_OVERLAP", "0") == "1" and not torch.is_grad_enabled()
            and torch.cuda.current_stream() != torch.cuda.default_stream()
            and not torch.cuda.is_current_stream_capturing())


def side_stream(dev=None):
    """The CU-masked stream of `dev` (created once per device and share); None when the runtime refuses a mask."""
    import ctypes
    import os
    dev = torch.cuda.current_device() if dev is None else torch.device(dev).index
    share = float(os.environ.get("DISGAT_SIDE_CUS", "0.75"))
    key = (dev, share)
    if key in _SIDE:
        return _SIDE[key]
    stream = None
    try:
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        words = (n_cu + 31) // 32
        mask = (ctypes.c_uint32 * words)()
        period = 8
        keep = max(1, min(period, round(share * period)))
        # the CUs left out are spread evenly whether CU ids run XCD by XCD or round-robin over the XCDs
        for i in range(n_cu):
            if ((i // 8) + i) % period < keep:
                mask[i // 32] |= 1 << (i % 32)
        hip = ctypes.CDLL("libamdhip64.so")
        ptr = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(ptr), ctypes.c_uint32(words), mask)
        if rc == 0 and ptr.value:
            stream = torch.cuda.ExternalStream(ptr.value, device=dev)
    except (OSError, AttributeError, RuntimeError):
        stream = None
    _SIDE[key] = stream
    return stream


def run_on_side(fn, inputs):
    """fn() on the side stream once everything the current stream has queued so far is done (its inputs are ready);
    `inputs`: the tensors fn reads (kept from reuse by the allocator until the side stream is through with them).  Returns
    fn's result; its tensors may be READ on the current stream only after join_side().  Falls back to a plain call."""
    side = side_stream()
    if side is None:
        return fn()
    main = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(main)
    side.wait_event(ready)
    for t in inputs:
        if t is not None:
            t.record_stream(side)
    with torch.cuda.stream(side):
        out = fn()
        done = torch.cuda.Event()
        done.record(side)
    _SIDE_PENDING.append((done, main))
    return out


def mark_side_outputs(tensors):
    """Outputs of run_on_side that the current stream will read after join_side(): allocated on the side stream's pool."""
    main = torch.cuda.current_stream()
    for t in tensors:
        if t is not None:
            t.record_stream(main)


def join_side():
    """The current stream waits for everything run_on_side has queued."""
    cur = torch.cuda.current_stream()
    while _SIDE_PENDING:
        done, _main = _SIDE_PENDING.pop()
        cur.wait_event(done)


def _check(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the DISGAT HIP path needs device tensors (got {t.device}); no CPU fallback exists")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")


def _row_major(t, name):
    if t.stride(-1) != 1:
        raise RuntimeError(f"{name}: innermost stride must be 1")
    return t


def edge_forward(graph: CSRGraph, att, H, F_in, F_out, x, rowop, colop, a, sage_div, drop=(0.0, 0), need_den=True,
                 sign=None, e_in=None, z_bound=None):
    """Launch disgat_edge_fwd (+ disgat_edge_combine for split rows).

    x [N, >=F_in] (row stride % 4 == 0), rowop/colop 2-D views with unit inner
    stride, a [H*F_out] or None; drop = (p, seed[, device uint64 counter added to the seed]) of the attention dropout; sign: optional
    sign_record(...) buffer the att-3 kernel fills; e_in: optional contiguous [H,E] partial scores added before the
    sigmoid.  z_bound: optional device scalar >= max |Z|: Z is then returned as ops_gemm.Planes (the two fp16 planes the
    f16x3 projection GEMM consumes, [N,H,F_in] each) instead of an fp32 tensor.
    Returns Z [N,H,F_in], edge_e [H,E], den [N,2,H].
    """
    if e_in is not None:
        _check(e_in, "e_in")
        if tuple(e_in.shape) != (H, graph.nnz) or not e_in.is_contiguous():
            raise RuntimeError("e_in must be a contiguous [H,E] tensor")
    for t, nm in ((x, "x"), (rowop, "rowop")) + (((colop, "colop"),) if colop is not None else ()):
        _check(t, nm)
        _row_major(t, nm)
    n, e = graph.n, graph.nnz
    dev = x.device
    wi = graph.work_items(chunk_for(att, graph, H))
    z = zh = zl = None
    if z_bound is None:
        z = torch.empty((n, H, F_in), dtype=torch.float32, device=dev)
    else:
        zh = torch.empty((n, H, F_in), dtype=torch.int16, device=dev)
        zl = torch.empty_like(zh)
    edge_e = torch.empty((H, e), dtype=torch.float32, device=dev)
    den = torch.empty((n, 2, H), dtype=torch.float32, device=dev) if need_den else None
    part_z = part_den = None
    if wi.n_slots:
        part_z = torch.empty((wi.n_slots, H, F_in), dtype=torch.float32, device=dev)
        part_den = torch.empty((wi.n_slots, 2, H), dtype=torch.float32, device=dev)
    st = _stream()
    _launch("disgat_edge_fwd", f"edge_fwd_att{att}", edge_algorithmic_bytes(att, n, e, H, F_in, F_out),
            att, _ptr(wi.items), wi.n_items, _ptr(graph.col), e, n, H, F_in, F_out,
              _ptr(x), x.stride(0), _ptr(rowop), rowop.stride(0), _ptr(colop), 0 if colop is None else colop.stride(0),
              _ptr(a), _ptr(z), _ptr(edge_e), _ptr(den), _ptr(part_z), _ptr(part_den), int(bool(sage_div)),
              float(drop[0]), int(drop[1]), _ptr(drop[2] if len(drop) > 2 else None), _ptr(sign), _ptr(e_in), _ptr(zh), _ptr(zl), _ptr(z_bound), st)
    if wi.n_split:
        _lib.call("disgat_edge_combine", _ptr(wi.split_rows), _ptr(wi.split_ptr), wi.n_split, H, F_in,
                  _ptr(part_z), _ptr(part_den), _ptr(z), _ptr(den), int(bool(sage_div)), _ptr(zh), _ptr(zl), _ptr(z_bound), st)
    if z_bound is not None:
        from .ops_gemm import Planes
        z = Planes(zh, zl, z_bound)
    return z, edge_e, den


# The kernels index the operand tables with the ids they are given: an out-of-range pair id is a GPU fault, not an
# exception.  With CHECK_INDICES on, every pair list is range-checked once on the host side (one aminmax per tensor
# OBJECT - the verdict is remembered on the tensor, so the second layer and later steps on the same list do not
# synchronise again; lists produced by sampling.sample_pairs are in range by construction and arrive pre-marked).
CHECK_INDICES = True


def mark_checked(pairs, n_rows=None, n_cols=None):
    """Record that `pairs` lies in [0, n_rows) x [0, n_cols) (None = "any table", for lists built from the graph's own
    ids) at its current version: check_pairs skips the device round trip for the same or a larger table while the
    tensor is not modified in place."""
    pairs._disgat_checked = (float("inf") if n_rows is None else int(n_rows), float("inf") if n_cols is None else int(n_cols),
                             pairs._version)
    return pairs


def check_pairs(pairs, n_rows, n_cols):
    """Range check of a pair list against the operand tables it will index.  The verdict is remembered ON the tensor
    together with the bounds it was checked against and the tensor's version counter: a list checked for a full graph
    is re-checked when it meets a smaller table (a row shard, a halo-compact table) or after an in-place edit."""
    if not CHECK_INDICES or pairs.shape[1] == 0:
        return
    seen = getattr(pairs, "_disgat_checked", None)
    if seen is not None and seen[2] == pairs._version and seen[0] <= n_rows and seen[1] <= n_cols:
        return
    r_lo, r_hi = torch.aminmax(pairs[0])
    c_lo, c_hi = torch.aminmax(pairs[1])
    r_lo, r_hi, c_lo, c_hi = torch.stack([r_lo, r_hi, c_lo, c_hi]).tolist()
    if r_lo < 0 or c_lo < 0 or r_hi >= n_rows or c_hi >= n_cols:
        raise RuntimeError(f"auxiliary pair list out of range: rows in [{r_lo},{r_hi}] (must be < {n_rows}), "
                           f"columns in [{c_lo},{c_hi}] (must be < {n_cols})")
    try:
        pairs._disgat_checked = (int(r_hi) + 1, int(c_hi) + 1, pairs._version)      # the tightest table this list fits
    except AttributeError:
        pass


def aux_forward(att, H, F_in, F_out, pairs, n, x, rowop, colop, a, h_lo=0, h_hi=None, sign=None):
    """Launch disgat_aux_score.  pairs: int64 [2,M] device tensor.  Returns [H,M]
    (rows outside [h_lo,h_hi) are left uninitialised and must not be read)."""
    if pairs.dtype != torch.int64 or not pairs.is_cuda or pairs.dim() != 2 or pairs.shape[0] != 2:
        raise RuntimeError("auxiliary pair list must be an int64 device tensor of shape (2,M)")
    h_hi = H if h_hi is None else h_hi
    # row ids index rowop (the rows this process owns); column ids index colop, or x for att 2 (every node)
    check_pairs(pairs, rowop.shape[0], (colop if colop is not None else x).shape[0])
    pairs = pairs.contiguous()
    m = int(pairs.shape[1])
    out = torch.empty((H, m), dtype=torch.float32, device=pairs.device)
    _launch("disgat_aux_score", f"aux_score_att{att}", aux_algorithmic_bytes(att, n, m, h_hi - h_lo, F_in, F_out),
            att, pairs[0].data_ptr(), pairs[1].data_ptr(), m, n, H, F_in, F_out, h_lo, h_hi,
              _ptr(x), 0 if x is None else x.stride(0), _ptr(rowop), rowop.stride(0),
              _ptr(colop), 0 if colop is None else colop.stride(0), _ptr(a), _ptr(out), _ptr(sign), _stream())
    return out


def pair_loss_sums(aux, h_lo, h_hi, labels, count=None, want_value=False):
    """disgat_pair_loss: a float64 device tensor [sum_sq_pos, sum_sq_neg, n_pos]; with want_value also (loss32, value):
    the list's utils.adj_mse_loss as a 0-d fp32 tensor and float64 {loss, neg_w, m}, finished inside the same launches
    (m = `count`, a 0-d float64 device tensor - the valid length of a fixed-capacity list - or the list length)."""
    _check(aux, "aux")
    _check(labels, "labels")
    if aux.stride(1) != 1 or aux.stride(0) != aux.shape[1]:
        raise RuntimeError("aux must be a contiguous [H,M] tensor")
    if count is not None and (count.dtype != torch.float64 or not count.is_cuda or count.numel() != 1):
        raise RuntimeError("pair_loss: count must be one float64 on the device")
    acc = torch.empty(3, dtype=torch.float64, device=aux.device)
    part = torch.empty((2048, 3), dtype=torch.float64, device=aux.device)      # DISGAT_PAIR_LOSS_MAX_BLOCKS per-block sums
    value = torch.empty(3, dtype=torch.float64, device=aux.device) if want_value else None
    loss32 = torch.empty((), dtype=torch.float32, device=aux.device) if want_value else None
    _lib.call("disgat_pair_loss", _ptr(aux), int(aux.shape[1]), h_lo, h_hi, _ptr(labels.contiguous()), _ptr(count), _ptr(acc),
              _ptr(part), _ptr(value), _ptr(loss32), _stream())
    return (acc, loss32, value) if want_value else acc


class ClsLoss(torch.autograd.Function):
    """disgat_cls_loss: log_softmax + NLL + accuracy of the rows of two splits in one pass over `logits` [N, C].

    code: int32 [N] row codes (-1 = in no split, else label + (split << 16)), or None with label_mod = the DifHead form
    (every row in split 0, label = row % label_mod).  div0 / div1: the splits' (global) sizes.
    Returns (loss, logp, res): loss = a 0-d fp32 tensor, split 0's NLL sum / div0 - the only differentiable output;
    logp [N, C]; res float64 [4] = (NLL_0 / div0, correct_0 / div0, NLL_1 / div1, correct_1 / div1)."""

    @staticmethod
    def forward(ctx, logits, code, label_mod, div0, div1):
        _check(logits, "logits")
        if logits.dim() != 2 or logits.stride(1) != 1:
            raise RuntimeError("cls_loss: logits must be [N, C] with unit inner stride")
        n, c = logits.shape
        if code is not None and (code.dtype != torch.int32 or not code.is_cuda or code.shape != (n,) or not code.is_contiguous()):
            raise RuntimeError("cls_loss: row codes must be a contiguous int32 device tensor [N]")
        logp = torch.empty((n, c), dtype=torch.float32, device=logits.device)
        res = torch.empty(4, dtype=torch.float64, device=logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        part = torch.empty((1024, 4), dtype=torch.float64, device=logits.device) if n > 8192 else None
        _lib.call("disgat_cls_loss", logits.data_ptr(), logits.stride(0), _ptr(code), int(label_mod), n, c, float(div0), float(div1),
                  logp.data_ptr(), logp.stride(0), _ptr(part), loss.data_ptr(), res.data_ptr(), _stream())
        ctx.save_for_backward(logp, code)
        ctx.cfg = (int(label_mod), float(div0))
        ctx.mark_non_differentiable(logp, res)
        ctx.set_materialize_grads(False)      # no zero tensors for the two outputs nobody differentiates
        return loss, logp, res

    @staticmethod
    def backward(ctx, g, _g_logp, _g_res):
        if g is None:
            return None, None, None, None, None
        logp, code = ctx.saved_tensors
        label_mod, div0 = ctx.cfg
        n, c = logp.shape
        gx = torch.empty_like(logp)
        g = g.reshape(1).contiguous().float()
        _lib.call("disgat_cls_loss_bwd", logp.data_ptr(), logp.stride(0), _ptr(code), label_mod, n, c, g.data_ptr(), div0,
                  gx.data_ptr(), gx.stride(0), _stream())
        return gx, None, None, None, None


def cls_loss(logits, code, label_mod, div0, div1=1.0):
    return ClsLoss.apply(logits, code, label_mod, div0, div1)


class EdgePass(torch.autograd.Function):
    """Differentiable wrapper of edge_forward.  Non-tensor config travels in `cfg`; e_in: optional [H,E] partial
    scores (its gradient is the total score gradient)."""

    @staticmethod
    def forward(ctx, x, rowop, colop, a, cfg, e_in=None):
        graph, att, H, F_in, F_out, sage, drop = cfg[:7]
        ctx.sign = sign_record(att, H, F_out, graph.nnz, x.device) if (len(cfg) > 7 and cfg[7] and graph.nnz) else None
        z, edge_e, den = edge_forward(graph, att, H, F_in, F_out, x, rowop, colop, a, sage, drop, sign=ctx.sign, e_in=e_in)
        ctx.cfg = cfg[:7]
        ctx.has_e_in = e_in is not None
        ctx.save_for_backward(x, rowop, colop, a, z, edge_e, den)
        ctx.mark_non_differentiable(den)
        return z, edge_e, den

    @staticmethod
    def backward(ctx, gz, ge, _gden):
        from . import ops_bwd
        out = ops_bwd.edge_backward(ctx, gz, ge, want_ge=ctx.has_e_in and ctx.needs_input_grad[5])
        return out[:5] + (out[5],)


class AuxPass(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rowop, colop, a, pairs, cfg):
        att, H, F_in, F_out, n, h_lo, h_hi = cfg[:7]
        m = int(pairs.shape[1])
        ctx.sign = sign_record(att, H, F_out, m, pairs.device) if (len(cfg) > 7 and cfg[7] and m) else None
        out = aux_forward(att, H, F_in, F_out, pairs, n, x, rowop, colop, a, h_lo, h_hi, sign=ctx.sign)
        ctx.cfg = cfg[:7]
        ctx.save_for_backward(x, rowop, colop, a, pairs)
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import ops_bwd
        return ops_bwd.aux_backward(ctx, gout)


class LayerPass(torch.autograd.Function):
    """Edge pass + every aux list of one att-3 layer as ONE autograd node (differentiable forwards with a sign
    record): the score operands P, Q and `a` then receive one gradient each, accumulated list after list inside the
    backward kernels, instead of one full [N, H*F_out] tensor per consumer summed by autograd.
    cfg = (graph, att, H, F_in, F_out, sage, drop, ranges) with ranges[i] = (h_lo, h_hi) of aux list i; the tensors after
    cfg are the aux lists.  With a ninth cfg entry (am_p, am_q) the node also OWNS the GEMMs behind P and Q: four more
    tensors follow the lists - x_p, W_top, x_q, W_bot with rowop = x_p W_top, colop = x_q W_bot - rowop / colop are taken
    as plain values (detached), nothing of them is saved, and the backward hands gradients to those four instead
    (ops_bwd.layer_backward_u: no operand table is read or rebuilt).  A tenth entry True detaches the edge list: its outputs
    are plain values (no sign record, non-differentiable) - layers.disga_heads(heads_discarded=True)."""

    @staticmethod
    def forward(ctx, x, rowop, colop, a, cfg, *rest):
        graph, att, H, F_in, F_out, sage, drop, ranges = cfg[:8]
        ctx.u_am = cfg[8] if len(cfg) > 8 else None
        edge_detached = len(cfg) > 9 and bool(cfg[9])     # the edge list's outputs are plain values: no sign record, no gradient
        lists = rest[:len(ranges)]
        ctx.sign = sign_record(att, H, F_out, graph.nnz, x.device) if (graph.nnz and not edge_detached) else None
        z, edge_e, den = edge_forward(graph, att, H, F_in, F_out, x, rowop, colop, a, sage, drop, sign=ctx.sign)
        outs, ctx.aux_signs = [], []
        for pairs, (lo, hi) in zip(lists, ranges):
            m = int(pairs.shape[1])
            sg = sign_record(att, H, F_out, m, pairs.device) if m else None
            outs.append(aux_forward(att, H, F_in, F_out, pairs, graph.n, None, rowop, colop, a, lo, hi, sign=sg))
            ctx.aux_signs.append(sg)
        ctx.cfg = cfg[:8]
        if ctx.u_am is not None:
            ctx.save_for_backward(x, a, z, edge_e, den, *lists, *rest[len(ranges):])
        else:
            ctx.save_for_backward(x, rowop, colop, a, z, edge_e, den, *lists)
        if edge_detached:
            ctx.mark_non_differentiable(z, edge_e, den)
        else:
            ctx.mark_non_differentiable(den)
        ctx.set_materialize_grads(False)      # outputs nobody differentiates arrive as None, and their passes are skipped
        return (z, edge_e, den, *outs)

    @staticmethod
    def backward(ctx, gz, ge, _gden, *gaux):
        from . import ops_bwd
        if ctx.u_am is not None:
            return ops_bwd.layer_backward_u(ctx, gz, ge, gaux)
        return ops_bwd.layer_backward(ctx, gz, ge, gaux)
