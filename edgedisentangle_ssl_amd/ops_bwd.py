"""Host orchestration of the backward passes (csrc/edge_bwd.hip) for ops.EdgePass / ops.AuxPass."""
import os

import torch

from . import _lib, ops
from .graph import build_items

_MAX_WAVES = 8192      # persistent seg_grad_att3 launch: 2048 blocks x 4 waves


def _buf(shape, dev, zero):
    return (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=dev)


# Keys split across several work items (hub rows / columns): True = every slice writes its own partial record and
# disgat_seg_combine adds them in slice order (run-to-run deterministic gradients); False = float atomics into zeroed
# rows (the round-1 behaviour, kept for comparison).
DETERMINISTIC = os.environ.get("DISGAT_ATOMIC_BWD") != "1"


def _keybuf(shape, dev, wi, force_zero=False):
    """Output rows of a segment pass: whole keys are stored; split keys are written by the combine pass (or, with
    DETERMINISTIC off, accumulated with atomics into rows zeroed here - a handful of hub rows, not the whole buffer)."""
    if force_zero:
        return _buf(shape, dev, True)
    out = _buf(shape, dev, False)
    if wi.n_split > 0 and not DETERMINISTIC:
        out.index_fill_(0, wi.split_rows.long(), 0.0)
    return out


def _part(wi, gkey):
    """Partial-record buffer [n_slots, row stride of gkey] for the split keys of `wi`, or None."""
    if not (DETERMINISTIC and wi.n_split > 0):
        return None
    return torch.empty((wi.n_slots, gkey.stride(0)), dtype=torch.float32, device=gkey.device)


def _combine(wi, part, gkey, width, accumulate, amax_out=None):
    if part is not None:
        _lib.call("disgat_seg_combine", wi.split_rows.data_ptr(), wi.split_ptr.data_ptr(), wi.n_split, width,
                  part.data_ptr(), gkey.data_ptr(), gkey.stride(0), int(bool(accumulate)), ops._ptr(amax_out), ops._stream())


def _seg_att3(wi, other, perm, g, lo, hi, H, f_out, keyop, otherop, a, n_keys, want_ga):
    dev = g.device
    gkey = _keybuf((n_keys, H * f_out), dev, wi)
    n_waves = min(_MAX_WAVES, (wi.n_items + 3) // 4 * 4)
    ga_part = torch.empty((n_waves, H * f_out), dtype=torch.float32, device=dev) if want_ga else None
    part = _part(wi, gkey)
    _lib.call("disgat_seg_grad_att3", wi.items.data_ptr(), wi.n_items, other.data_ptr(), ops._ptr(perm), g.data_ptr(),
              g.stride(0), lo, hi, H, f_out, keyop.data_ptr(), keyop.stride(0), otherop.data_ptr(), otherop.stride(0),
              ops._ptr(a), gkey.data_ptr(), gkey.stride(0), ops._ptr(ga_part), n_waves, ops._ptr(part), ops._stream())
    _combine(wi, part, gkey, H * f_out, False)
    return gkey, (ga_part.sum(0) if want_ga else None)


def _g_strides(g, H):
    """(tensor to hand to the kernel, head stride, position stride) of a score gradient [H, M]: the [M, H]-backed layout
    disgat_pair_loss_bwd / disgat_bwd_alpha write for the sign path is taken as it is, anything else made contiguous."""
    if g.dim() == 2 and g.shape[0] == H and g.stride(0) == 1 and (g.stride(1) == H or g.shape[1] <= 1):
        return g, 1, H
    g = g.contiguous()
    return g, g.stride(0), 1


def _seg_sign(wi, perm, g, lo, hi, H, f_out, sign, keyop, a, n_keys, want_ga, into=None, amax_out=None):
    """Score backward of one side (rows: keyop = P, columns: keyop = Q) from the forward's sign record.
    Returns (gkey [n_keys, H*f_out], this side's share of grad a or None); `into`: an existing gkey to add into."""
    g, g_hs, g_ps = _g_strides(g, H)
    dev = g.device
    gkey = _keybuf((n_keys, H * f_out), dev, wi) if into is None else into
    n_waves = min(_MAX_WAVES, (wi.n_items + 3) // 4 * 4)
    ga_part = torch.empty((n_waves, H * f_out), dtype=torch.float32, device=dev) if want_ga else None
    part = _part(wi, gkey)
    _lib.call("disgat_seg_grad_sign", wi.items.data_ptr(), wi.n_items, ops._ptr(perm), g.data_ptr(), g_hs, g_ps, lo, hi,
              H, f_out, sign.data_ptr(), ops._ptr(keyop), 0 if keyop is None else keyop.stride(0), ops._ptr(a), gkey.data_ptr(), gkey.stride(0),
              ops._ptr(ga_part), n_waves, int(into is not None), ops._ptr(part),
              ops._ptr(amax_out if (part is not None or wi.n_split == 0) else None), ops._stream())
    _combine(wi, part, gkey, H * f_out, into is not None, amax_out)
    return gkey, (ga_part.sum(0) if want_ga else None)


def _seg_hx(col_mode, wi, other, perm, coef, lo, hi, H, f, otherop, gkey, accumulate):
    part = _part(wi, gkey)
    _lib.call("disgat_seg_grad_hx", int(col_mode), wi.items.data_ptr(), wi.n_items, other.data_ptr(), ops._ptr(perm),
              coef.data_ptr(), coef.stride(0), lo, hi, H, f, otherop.data_ptr(), otherop.stride(0), gkey.data_ptr(),
              gkey.stride(0), int(accumulate), ops._ptr(part), ops._stream())
    _combine(wi, part, gkey, f if col_mode else H * f, accumulate)


def _seg_sum(wi, perm, g, lo, hi, H, n_keys):
    """att 1: [n_keys, H] segment sums of the score gradients g [H, M] over a key-sorted list (fixed order, no atomics)."""
    g, g_hs, g_ps = _g_strides(g, H)
    ld = (H + 3) // 4 * 4
    gkey = _keybuf((n_keys, ld), g.device, wi)
    part = _part(wi, gkey)
    _lib.call("disgat_seg_sum", wi.items.data_ptr(), wi.n_items, ops._ptr(perm), g.data_ptr(), g_hs, g_ps, lo, hi, H,
              gkey.data_ptr(), ld, 0, ops._ptr(part), ops._stream())
    _combine(wi, part, gkey, ld, False)
    return gkey if ld == H else gkey[:, :H]


def edge_backward(ctx, gz, ge, want_ge=False):
    """Returns (g_x, g_rowop, g_colop, g_a, None, ge_tot or None): the last entry is the total gradient of the raw
    scores [H,E] (= the gradient of an additive e_in input), only when want_ge."""
    x, rowop, colop, a, z, edge_e, den = ctx.saved_tensors
    graph, att, H, f_in, f_out, sage, drop = ctx.cfg
    need_x, need_row, need_col, need_a = ctx.needs_input_grad[:4]
    dev = x.device
    n, e = graph.n, graph.nnz
    if e == 0:                      # no edges: Z == 0 and no score exists, every gradient is zero
        return None, None, None, None, None, None
    chunk = ops.chunk_for(att, graph, H)
    wi = graph.work_items(chunk)
    gz = torch.zeros_like(z) if gz is None else gz.contiguous()
    ge = None if ge is None else ge.contiguous()
    ge_tot = torch.empty((H, e), dtype=torch.float32, device=dev)
    beta = torch.empty((H, e), dtype=torch.float32, device=dev)
    _lib.call("disgat_bwd_alpha", wi.items.data_ptr(), wi.n_items, graph.col.data_ptr(), e, H, f_in, x.data_ptr(),
              x.stride(0), gz.data_ptr(), z.data_ptr(), edge_e.data_ptr(), den.data_ptr(), ops._ptr(ge), ge_tot.data_ptr(), 0,
              beta.data_ptr(), int(bool(sage)), float(drop[0]), int(drop[1]), ops._ptr(drop[2] if len(drop) > 2 else None), ops._stream())
    g_x = g_row = g_col = g_a = None
    t = twi = None
    sign = getattr(ctx, "sign", None)
    if need_x or (att in (1, 4) and need_col) or (att == 3 and (need_col or (need_a and sign is not None))):
        t = graph.transpose()
        twi = t.work_items(chunk)
    if att == 1 and DETERMINISTIC:                      # gs1[r] = sum over the row's edges, gs2[c] = sum over the column's
        if need_row:
            g_row = _seg_sum(wi, None, ge_tot, 0, H, H, n)
        if need_col:
            g_col = _seg_sum(twi, t.eid, ge_tot, 0, H, H, colop.shape[0])
    elif att == 1:
        if need_row:
            g_row = torch.zeros((n, H), dtype=torch.float32, device=dev).index_add_(0, graph.row, ge_tot.t())
        if need_col:
            g_col = torch.zeros((colop.shape[0], H), dtype=torch.float32, device=dev).index_add_(
                0, graph.col.long(), ge_tot.t())
    elif att == 2:
        if need_row:                                    # gP[r,h,:] = sum_k ge_k x[col_k]
            g_row = _keybuf((n, H * f_in), dev, wi)
            _seg_hx(0, wi, graph.col, None, ge_tot, 0, H, H, f_in, x, g_row, False)
    elif att == 4:                                      # e = <h[r], h[c]> per head: gP[r] = sum ge h[c], gQ[c] = sum ge h[r]
        if need_row:
            g_row, _ = _seg_att3(wi, graph.col, None, ge_tot, 0, H, H, f_out, rowop, colop, None, n, False)
        if need_col:
            g_col, _ = _seg_att3(twi, t.col, t.eid, ge_tot, 0, H, H, f_out, colop, rowop, None, colop.shape[0], False)
    elif sign is not None:                              # gather-free: both sides read the sign record
        if need_row or need_a:
            g_row, g_a = _seg_sign(wi, None, ge_tot, 0, H, H, f_out, sign, rowop, a, n, need_a)
        if need_col or need_a:
            g_col, ga_c = _seg_sign(twi, t.eid, ge_tot, 0, H, H, f_out, sign, colop, a, colop.shape[0], need_a)
            g_a = g_a + ga_c if need_a else None
    else:
        if need_row or need_a:
            g_row, g_a = _seg_att3(wi, graph.col, None, ge_tot, 0, H, H, f_out, rowop, colop, a, n, need_a)
        if need_col:
            g_col, _ = _seg_att3(twi, t.col, t.eid, ge_tot, 0, H, H, f_out, colop, rowop, a, colop.shape[0], False)
    if need_x:
        g_x = _keybuf(tuple(x.shape), dev, twi, x.stride(0) != f_in)
        # grad of the aggregation: gx[c] = sum_{k in col c} sum_h beta_kh gZ[row_k,h,:]
        _seg_hx(1, twi, t.col, t.eid, beta, 0, H, H, f_in, gz.view(n, H * f_in), g_x, False)   # n rows of gZ, g_x over all columns
        if att == 2:                                    # e = <P[r,h,:], x[c,:]>  ->  gx[c] += sum_h ge_kh P[r_k,h,:]
            _seg_hx(1, twi, t.col, t.eid, ge_tot, 0, H, H, f_in, rowop, g_x, True)
    return g_x, g_row, g_col, g_a, None, (ge_tot if want_ge else None)


_SEG_CACHE = {}          # (storage ptr, M, version, side, n_keys, chunk) -> (pairs kept alive, result); FIFO of 8


def clear_segment_cache():
    """Drop the memoised sorts / work items (each entry pins its pair list, an int32 permutation and the items: at
    1M nodes several GB).  Pair lists are resampled every step, so Trainer._finish_step calls this after backward."""
    _SEG_CACHE.clear()


def _segments_of(pairs, side, n_keys, chunk):
    """_segments(pairs[side]) memoised per pair list: the same list is scored by both layers of a pass, so its
    column-side sort (the expensive part) and work items are built once per training step instead of once per layer.
    The entry keeps `pairs` alive, so its storage pointer cannot be recycled while the entry exists."""
    key = (pairs.data_ptr(), int(pairs.shape[1]), pairs._version, side, n_keys, chunk)
    hit = _SEG_CACHE.get(key)
    if hit is not None:
        return hit[1]
    if getattr(pairs, "_disgat_static", False):
        res = _segments_static(pairs[side], n_keys, side == 0, chunk)
    else:
        res = _segments(pairs[side], n_keys, chunk)
    if len(_SEG_CACHE) >= 8:
        _SEG_CACHE.pop(next(iter(_SEG_CACHE)))
    _SEG_CACHE[key] = (pairs, res)
    return res


def _segments(keys, n_keys, chunk):
    """Work items over a key-sorted list + the int32 permutation that sorts it (None if sorted)."""
    m = keys.numel()
    if m > 1 and not bool((keys[1:] >= keys[:-1]).all()):
        # node ids fit 32 bits: the radix sort then runs 4 digit passes instead of 8 (66M column ids: 2.2 -> 1.1 ms)
        perm = torch.sort(keys.to(torch.int32) if n_keys < 2 ** 31 else keys, stable=True).indices
        keys = keys[perm]
        perm32 = perm.to(torch.int32)
    else:
        perm, perm32 = None, None
    ptr = torch.searchsorted(keys, torch.arange(n_keys + 1, device=keys.device, dtype=keys.dtype))
    return build_items(ptr, chunk), perm, perm32


def _segments_static(keys, n_keys, is_sorted, chunk):
    """_segments() for the fixed-capacity lists of captured steps (sampling.PairSampler.sample_static): no size is read back
    and every shape is fixed.  Keys longer than `chunk` entries are cut into near-equal slices exactly as build_items() cuts
    them, but the item table has a fixed CAPACITY - n_keys + C // chunk bounds sum_k max(1, ceil(deg_k / chunk)) - and the
    entries past the real count are padding (key -1: the kernels skip them), likewise the split-key table (at most
    C // chunk keys are longer than chunk).  Items stay in key order.  Three launches of this library
    (disgat_seg_tables: csrc/seg_tables.hip) behind the column side's sort."""
    from .graph import WorkItems
    dev = keys.device
    c_len = int(keys.numel())
    perm = None if is_sorted else torch.sort(keys.to(torch.int32) if n_keys < 2 ** 31 else keys, stable=True).indices
    slices = DETERMINISTIC and chunk < c_len
    eff = chunk if slices else max(chunk, c_len, 1)              # no slices: one item per key
    cap = n_keys + c_len // eff
    n_split_cap = max(1, c_len // eff)
    i32 = dict(dtype=torch.int32, device=dev)
    ptr, key_off = torch.empty(n_keys + 1, **i32), torch.empty(2 * n_keys, **i32)
    perm32 = None if perm is None else torch.empty(c_len, **i32)
    wi = WorkItems()
    wi.chunk = chunk
    wi.items = torch.empty((cap, 4), **i32)
    wi.split_rows, wi.split_ptr = torch.empty(n_split_cap, **i32), torch.empty(n_split_cap + 1, **i32)
    wi.totals = torch.empty(4, **i32)                            # items, split keys, slots actually in use (device)
    _lib.call("disgat_seg_tables", keys.data_ptr(), ops._ptr(perm), c_len, int(n_keys), int(eff), ptr.data_ptr(), key_off.data_ptr(),
              ops._ptr(perm32), wi.items.data_ptr(), cap, wi.split_rows.data_ptr(), wi.split_ptr.data_ptr(), n_split_cap,
              wi.totals.data_ptr(), ops._stream())
    wi.n_items = int(cap)
    wi.n_split = int(n_split_cap) if slices else 0
    wi.n_slots = int(2 * (c_len // chunk) + 2) if slices else 0   # sum over keys longer than chunk of ceil(deg / chunk) <= 2 C / chunk
    return wi, perm, perm32


def aux_backward(ctx, gout):
    x, rowop, colop, a, pairs = ctx.saved_tensors
    att, H, f_in, f_out, n, lo, hi = ctx.cfg
    need_x, need_row, need_col, need_a = ctx.needs_input_grad[:4]
    dev = gout.device
    if pairs.shape[1] == 0:         # empty pair list: nothing was scored
        return None, None, None, None, None, None
    gout = gout.contiguous()
    rows, cols = pairs[0], pairs[1]
    g_x = g_row = g_col = g_a = None
    if att == 1 and DETERMINISTIC:
        chunk = ops.chunk_small(att, int(pairs.shape[1]), H)
        if need_row:
            wi, _perm, perm32 = _segments_of(pairs, 0, rowop.shape[0], chunk)
            g_row = _seg_sum(wi, perm32, gout, lo, hi, H, rowop.shape[0])
        if need_col:
            wi, _perm, perm32 = _segments_of(pairs, 1, colop.shape[0], chunk)
            g_col = _seg_sum(wi, perm32, gout, lo, hi, H, colop.shape[0])
        return g_x, g_row, g_col, g_a, None, None
    if att == 1:
        gsub = gout[lo:hi].t()
        if need_row:
            g_row = torch.zeros_like(rowop)
            g_row[:, lo:hi] = torch.zeros((rowop.shape[0], hi - lo), dtype=torch.float32, device=dev).index_add_(0, rows, gsub)
        if need_col:
            g_col = torch.zeros_like(colop)
            g_col[:, lo:hi] = torch.zeros((colop.shape[0], hi - lo), dtype=torch.float32, device=dev).index_add_(0, cols, gsub)
        return g_x, g_row, g_col, g_a, None, None
    chunk = ops.chunk_small(att, int(pairs.shape[1]), H)
    n_rows = rowop.shape[0]
    n_cols = colop.shape[0] if att in (3, 4) else x.shape[0]
    sign = getattr(ctx, "sign", None)
    if att == 3 and sign is not None:                   # gather-free: only the keys and the sign record are read
        if need_row or need_a:
            wi, _perm, perm32 = _segments_of(pairs, 0, n_rows, chunk)
            g_row, g_a = _seg_sign(wi, perm32, gout, lo, hi, H, f_out, sign, rowop, a, n_rows, need_a)
        if need_col or need_a:
            wi, _perm, perm32 = _segments_of(pairs, 1, n_cols, chunk)
            g_col, ga_c = _seg_sign(wi, perm32, gout, lo, hi, H, f_out, sign, colop, a, n_cols, need_a)
            g_a = g_a + ga_c if need_a else None
        return g_x, g_row, g_col, g_a, None, None
    if need_row or need_a:
        wi, perm, perm32 = _segments_of(pairs, 0, n_rows, chunk)
        other = (cols if perm is None else cols[perm]).to(torch.int32)
        if att == 4:
            g_row, _ = _seg_att3(wi, other, perm32, gout, lo, hi, H, f_out, rowop, colop, None, n_rows, False)
        elif att == 3:
            g_row, g_a = _seg_att3(wi, other, perm32, gout, lo, hi, H, f_out, rowop, colop, a, n_rows, need_a)
        else:
            g_row = _keybuf((n_rows, H * f_in), dev, wi)
            _seg_hx(0, wi, other, perm32, gout, lo, hi, H, f_in, x, g_row, False)
    if (att in (3, 4) and need_col) or (att == 2 and need_x):
        wi, perm, perm32 = _segments_of(pairs, 1, n_cols, chunk)
        other = (rows if perm is None else rows[perm]).to(torch.int32)
        if att in (3, 4):
            g_col, _ = _seg_att3(wi, other, perm32, gout, lo, hi, H, f_out, colop, rowop, a if att == 3 else None, n_cols, False)
        else:
            g_x = _keybuf(tuple(x.shape), dev, wi, x.stride(0) != f_in)
            _seg_hx(1, wi, other, perm32, gout, lo, hi, H, f_in, rowop, g_x, False)
    return g_x, g_row, g_col, g_a, None, None


def layer_backward_u(ctx, gz, ge, gaux):
    """layer_backward for a pass that owns the GEMMs of its score operands (ops.LayerPass with cfg[8]; att 3 with sign
    records): P = x_p W_top and Q = x_q W_bot are neither saved nor rebuilt.  The segment passes return u = sum g lrelu'
    per key WITHOUT the a-scale (disgat_seg_grad_sign, a == NULL), and with G = x^T u - the weight-gradient GEMM that
    runs anyway -   grad W = G (.) a,   grad a = sum_rows W (.) G  (= sum_key (x W)[key] (.) u[key]),   grad x = u (W (.) a)^T.
    No pass reads an operand table; the row side is finished (and its 8 GB u freed) before the column side starts; the
    sign records go list by list during the column phase."""
    from . import ops_gemm
    graph, att, H, f_in, f_out, sage, drop, ranges = ctx.cfg
    n_lists = len(ranges)
    x, a, z, edge_e, den, *rest = ctx.saved_tensors
    lists, (x_p, w_top, x_q, w_bot) = rest[:n_lists], rest[n_lists:]
    am_p, am_q = ctx.u_am
    need = ctx.needs_input_grad
    need_x, need_a = need[0], need[3]
    need_xp, need_wt, need_xq, need_wb = need[5 + n_lists: 9 + n_lists]
    dev = x.device
    n, e = graph.n, graph.nnz
    chunk = ops.chunk_for(att, graph, H)
    n_rows, n_cols = int(x_p.shape[0]), int(x_q.shape[0])
    have_edge = (gz is not None or ge is not None) and e > 0
    ge_tot = beta = t = twi = None
    if have_edge:
        wi = graph.work_items(chunk)
        gz = torch.zeros_like(z) if gz is None else gz.contiguous()
        ge = None if ge is None else ge.contiguous()
        ge_tot = torch.empty((e, H), dtype=torch.float32, device=dev).t()     # [H, E] backed by [E, H]: both segment passes
        beta = torch.empty((H, e), dtype=torch.float32, device=dev)            # read a position's H gradients as one 32-byte run
        _lib.call("disgat_bwd_alpha", wi.items.data_ptr(), wi.n_items, graph.col.data_ptr(), e, H, f_in, x.data_ptr(),
                  x.stride(0), gz.data_ptr(), z.data_ptr(), edge_e.data_ptr(), den.data_ptr(), ops._ptr(ge), ge_tot.data_ptr(), 1,
                  beta.data_ptr(), int(bool(sage)), float(drop[0]), int(drop[1]), ops._ptr(drop[2] if len(drop) > 2 else None), ops._stream())
        t = graph.transpose()
        twi = t.work_items(chunk)
    live = [(pairs, rng, sg, go) for pairs, rng, sg, go in zip(lists, ranges, ctx.aux_signs, gaux)
            if go is not None and sg is not None]          # [H, M], possibly [M, H]-backed (_g_strides)
    g_a = None

    def dense_side(x_in, w, u, am, need_in, need_w, u_amax, add_to=None, with_a=True):
        """(grad x_in, grad w, this side's share of grad a) from u [rows, H*F_out]; u_amax: the bound on |u| the segment
        passes kept while storing it (max over everything they stored, so >= max |u|), or None; add_to: another
        contribution to grad x_in, added in the GEMM's epilogue; with_a = False: no weight-gradient GEMM for grad a's sake
        (a second call will run it)."""
        nonlocal g_a
        if u is None:
            return add_to, None
        want_a = need_a and with_a
        with torch.no_grad():
            g_in, G = ops_gemm.linear_backward(x_in.detach(), (w.detach() * a.detach()) if need_in else w.detach(), u, am,
                                               need_in, need_w or want_a, g_amax=u_amax, ga_init=add_to if need_in else None)
            g_w = None
            if G is not None:
                if want_a:
                    ga = (w.detach() * G).sum(0)
                    g_a = ga if g_a is None else g_a + ga
                if need_w:
                    g_w = G.mul_(a.detach())
        return g_in, g_w

    # ---- row side
    u = None
    bound = torch.zeros(2, dtype=torch.float32, device=dev) if DETERMINISTIC else None     # [row side, column side] max |u|
    b_row, b_col = (bound[0:1], bound[1:2]) if bound is not None else (None, None)
    if need_xp or need_wt or need_a:
        if have_edge:
            u, _ = _seg_sign(wi, None, ge_tot, 0, H, H, f_out, ctx.sign, None, None, n, False, amax_out=b_row)
        for pairs, (lo, hi), sg, gout in live:
            wl, _perm, perm32 = _segments_of(pairs, 0, n_rows, ops.chunk_small(att, int(pairs.shape[1]), H))
            u, _ = _seg_sign(wl, perm32, gout, lo, hi, H, f_out, sg, None, None, n_rows, False, into=u, amax_out=b_row)
    g_xp, g_wt = dense_side(x_p, w_top, u, am_p, need_xp, need_wt, b_row)
    u = None
    # ---- column side; every record is dead after its column pass
    if need_xq or need_wb or need_a:
        if have_edge:
            u, _ = _seg_sign(twi, t.eid, ge_tot, 0, H, H, f_out, ctx.sign, None, None, n_cols, False, amax_out=b_col)
        ctx.sign = None
        for li, (pairs, (lo, hi), sg, gout) in enumerate(live):
            wl, _perm, perm32 = _segments_of(pairs, 1, n_cols, ops.chunk_small(att, int(pairs.shape[1]), H))
            u, _ = _seg_sign(wl, perm32, gout, lo, hi, H, f_out, sg, None, None, n_cols, False, into=u, amax_out=b_col)
            live[li] = None
            sg = None
    ctx.sign = None
    ctx.aux_signs = [None] * len(ctx.aux_signs)
    # x_p, x_q (unsharded: the same tensor) and the aggregation input x (the same again unless it was padded) receive up to
    # three contributions to ONE gradient: the second GEMM adds the first's result in its epilogue and the aggregation pass
    # accumulates into that buffer, instead of autograd summing three [N, F_in] tensors afterwards
    same_pq = x_p is x_q and need_xp and need_xq and g_xp is not None
    # sharded: x_q is the gathered table and its gradient must travel back to the rows' owners (reduce-scatter).  Its data
    # gradient comes first, the aggregation's share is added, the reduce-scatter goes on the links (parallel.start_adjoint)
    # and the weight-gradient GEMM of the same operand runs under it; _AllGatherRows.backward then only collects the result
    early = need_xq and u is not None and getattr(x_q, "_disgat_gather", None) is not None
    if early:
        g_xq, _ = dense_side(x_q, w_bot, u, am_q, True, False, b_col, add_to=g_xp if same_pq else None, with_a=False)
    else:
        g_xq, g_wb = dense_side(x_q, w_bot, u, am_q, need_xq, need_wb, b_col, add_to=g_xp if same_pq else None)
    if same_pq:
        g_xp = None
    g_x = None
    if have_edge and need_x:
        tgt = g_xq if (x is x_q and g_xq is not None and g_xq.shape == x.shape and g_xq.stride(0) == f_in) else None
        if tgt is not None and DETERMINISTIC:
            _seg_hx(1, twi, t.col, t.eid, beta, 0, H, H, f_in, gz.view(n, H * f_in), tgt, True)
        else:
            g_x = _keybuf(tuple(x.shape), dev, twi, x.stride(0) != f_in)
            _seg_hx(1, twi, t.col, t.eid, beta, 0, H, H, f_in, gz.view(n, H * f_in), g_x, False)
    if early:
        from . import parallel
        if g_x is None:                  # g_xq is the whole gradient of the table (the aggregation added its share above)
            parallel.start_adjoint(x_q, g_xq)
        _, g_wb = dense_side(x_q, w_bot, u, am_q, False, need_wb, b_col)
    u = None
    return (g_x, None, None, g_a if need_a else None, None) + (None,) * n_lists + (g_xp, g_wt, g_xq, g_wb)


def layer_backward(ctx, gz, ge, gaux):
    """Backward of ops.LayerPass: the edge list first (stores the operand gradients), then every aux list ADDS its
    share into the same buffers inside seg_grad_sign_kernel."""
    from types import SimpleNamespace
    x, rowop, colop, a, z, edge_e, den, *lists = ctx.saved_tensors
    graph, att, H, f_in, f_out, sage, drop, ranges = ctx.cfg
    need_x, need_row, need_col, need_a = ctx.needs_input_grad[:4]
    g_x = g_row = g_col = g_a = None
    if (gz is not None or ge is not None) and graph.nnz:
        ectx = SimpleNamespace(saved_tensors=(x, rowop, colop, a, z, edge_e, den), cfg=(graph, att, H, f_in, f_out, sage, drop),
                               needs_input_grad=ctx.needs_input_grad, sign=ctx.sign)
        g_x, g_row, g_col, g_a, _, _ge = edge_backward(ectx, gz, ge)
    ctx.sign = None                    # 256 B per edge: the record is dead once both segment passes have read it
    n_rows, n_cols = rowop.shape[0], colop.shape[0]
    for li, (pairs, (lo, hi), gout) in enumerate(zip(lists, ranges, gaux)):
        sign, ctx.aux_signs[li] = ctx.aux_signs[li], None          # 256 B per pair (17 GB for a 66M-pair list): freed list by list
        if gout is None or sign is None:
            continue
        gout = gout.contiguous()
        chunk = ops.chunk_small(att, int(pairs.shape[1]), H)
        if need_row or need_a:
            wi, _perm, perm32 = _segments_of(pairs, 0, n_rows, chunk)
            g_row, ga = _seg_sign(wi, perm32, gout, lo, hi, H, f_out, sign, rowop, a, n_rows, need_a, into=g_row)
            if need_a:
                g_a = ga if g_a is None else g_a + ga
        if need_col or need_a:
            wi, _perm, perm32 = _segments_of(pairs, 1, n_cols, chunk)
            g_col, ga = _seg_sign(wi, perm32, gout, lo, hi, H, f_out, sign, colop, a, n_cols, need_a, into=g_col)
            if need_a:
                g_a = ga if g_a is None else g_a + ga
        del sign
    return (g_x, g_row if need_row else None, g_col if need_col else None, g_a if need_a else None, None) + (None,) * len(lists)
