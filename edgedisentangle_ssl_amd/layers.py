"""Host-side mirror of the reference's layer classes on the DISGAT path.

Same class names, constructor signatures, parameter names (= state_dict keys)
and initialisers as /root/reference/layers.py, so checkpoints and seeds carry
over; the sparse forward of all heads of a layer is ONE fused HIP edge pass
(csrc/edge_fwd.hip) plus dense GEMMs, instead of H sequential ATen pipelines.

  DisGALayer        layers.py:303-511   (one attention head; parameters only + single-head forward)
  SageConv          layers.py:63-112    (parameter container; math fused in disga_heads)
  GraphConvolution  layers.py:16-59     (parameter container; math fused in disga_heads)
  FuseLayer         layers.py:876-921
  disga_heads()     the H-head loop of models.py:225-228 / 240-243 as one call
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, ops_gemm, parallel
from .graph import graph_of


def _pow2ceil(v):
    p = 1
    while p < v:
        p *= 2
    return p


class GraphConvolution(nn.Module):
    """Parameters of the reference GCN layer (layers.py:16-36): weight [F_in,F_out], bias [F_out],
    uniform(+-1/sqrt(F_out))."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.FloatTensor(in_features, out_features))
        if bias:
            self.bias = nn.Parameter(torch.FloatTensor(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)


class SageConv(nn.Module):
    """Parameters of the reference GraphSage layer (layers.py:63-82): proj = Linear(2*F_in -> F_out,
    bias=False) with N(0,1) weights."""

    def __init__(self, in_features, out_features, bias=False):
        super().__init__()
        self.proj = nn.Linear(in_features * 2, out_features, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.proj.weight)
        if self.proj.bias is not None:
            nn.init.constant_(self.proj.bias, 0.0)


class DisGALayer(nn.Module):
    """One edge-disentangled attention head (layers.py:303-337 for the parameters).

    forward(input, adj, aux_indices=None) keeps the reference's single-head contract
    (layers.py:493-511): returns (elu(h'), edge_e[E,1]) or (elu(h'), edge_e, [aux_e[M,1]...]).
    DISGAT does not call it per head; it hands all heads of a layer to disga_heads().
    """

    def __init__(self, in_features, out_features, dropout, alpha, concat=True, att_type=1, gnn_type="AT"):
        super().__init__()
        self.dropout = dropout
        self.in_features = in_features
        self.out_features = out_features
        self.alpha = alpha            # unused by the reference as well (no LeakyReLU(alpha) on att 1)
        self.concat = concat
        self.att_type = att_type
        self.gnn_type = gnn_type
        if att_type == 3:
            self.W = nn.Parameter(torch.zeros(size=(in_features * 2, out_features)))
            nn.init.xavier_uniform_(self.W.data, gain=1.414)
            self.a = nn.Parameter(torch.zeros(size=(out_features, 1)))
            nn.init.xavier_uniform_(self.a.data, gain=1.414)
        else:
            self.W = nn.Parameter(torch.zeros(size=(in_features, out_features)))
            nn.init.xavier_uniform_(self.W.data, gain=1.414)
            self.a = nn.Parameter(torch.zeros(size=(2 * out_features, 1)))
            nn.init.xavier_uniform_(self.a.data, gain=1.414)
        if gnn_type == "AT":
            self.W_em = nn.Parameter(torch.zeros(size=(in_features, out_features)))
            nn.init.xavier_uniform_(self.W_em.data, gain=1.414)
        elif gnn_type == "SAGE":
            self.ag_layer = SageConv(in_features, out_features)
        elif gnn_type == "GCN":
            self.ag_layer = GraphConvolution(in_features, out_features)
        else:
            raise ValueError("not implemented for gnn_type {} in DISGAT".format(gnn_type))   # layers.py:409

    def forward(self, input, adj, aux_indices=None):
        heads, edge_e, aux = disga_heads([self], input, adj, aux_indices)
        out = heads[0] if self.concat else heads.pre_elu[0]
        if aux_indices is not None:
            if self.concat:
                return out, edge_e[0], aux[0]
            return out, edge_e[0]                      # layers.py:501-502 drops the aux scores
        return out, edge_e[0]


class HeadList(list):
    """List of the H per-head outputs [N,F_out] (what the reference's fusers receive,
    models.py:230-233).  All heads are column slices of ONE buffer `fused` [N, H*F_out], which a
    fuser may use directly instead of re-concatenating (FuseLayer below does); `fused_amax` is an upper
    bound of max |fused| (device scalar) for the GEMMs that consume it."""
    fused = None
    fused_amax = None
    pre_elu = None
    planes = None       # ops_gemm.Planes of the same [N, H*F_out] buffer (no-graph forwards): what FuseLayer / the DifHead
    n_heads = 0         # classifier GEMMs consume.  With `heads_planes` the fp32 buffer is never written and the list is
    f_out = 0           # EMPTY: only a consumer that asked for planes may receive such a list (DISGAT._run checks).
    deferred = None     # DeferredHeads: the projection has NOT run yet - a FuseLayer that can take it back to back
                        # (csrc/gemm_b2b.hip) runs projection + ELU + fuser as one launch and no head buffer ever exists;
                        # any other consumer calls deferred.materialize(heads) first (the plane GEMM, as before).


class DeferredHeads:
    """The per-head output projection of a no-graph layer pass, not yet run: Z planes [H, N, F_in], the stacked weights
    (layers.py:397-399 AT / :404-407 GCN), the optional per-head bias, the bound of the ELU output."""

    def __init__(self, zp, layers, gnn, w_rm, bias, h_bound, f_in, f_out):
        self.zp, self.layers, self.gnn, self.w_rm, self.bias, self.h_bound = zp, layers, gnn, w_rm, bias, h_bound
        self.H, self.f_in, self.f_out = zp.shape[0], f_in, f_out

    def w1_stack(self):
        return torch.stack([l.W_em if self.gnn == "AT" else l.ag_layer.weight for l in self.layers])

    def materialize(self, heads):
        """Run the projection as the plane GEMM (head buffer written as planes only) and hang the result on `heads`."""
        _act, hpl = ops_gemm.linear_planes(self.zp, self.w_rm, self.f_out, self.bias, None, ops_gemm.ACT_ELU, 0.01,
                                           want_f32=False, out_bound=self.h_bound)
        heads.planes, heads.deferred = hpl, None
        return hpl


MAX_HEAD_SLICE = 1024      # att 3 / 4: features of one head a single launch scores (2 heads x 32 lanes x 8 float4)


def _kernel_heads(att, Hp, f_out, f_in_p=0):
    """Heads per kernel launch (a power of two in [2,16]) and the padded per-head F_out for att 3.

    The att-3 lane map gives a head G = 64/Hk lanes x QN float4 (QN <= 8), i.e. Hk * F_out_padded <= 2048
    per launch; wider layers (more heads, wider heads) run as several head groups, each a launch over
    its slice of the operands."""
    hk = min(Hp, 16)
    if att == 2:                                    # the dot product over x cannot be sliced: fewer heads per
        while hk > 2 and 256 * (16 // hk) < min(f_in_p, 512):   # launch buy a wider register tile instead
            hk //= 2
    if att not in (3, 4):
        return hk, f_out
    while hk >= 2:
        g4 = (64 // hk) * 4
        qn = _pow2ceil((f_out + g4 - 1) // g4)
        if qn <= 8:
            return hk, qn * g4
        hk //= 2
    raise NotImplementedError(f"att={att} kernel envelope: a single head wider than 1024 features (nhid={f_out})")


def _memo(layers, tag, build):
    """Per-layer cache of packed weights (and their GEMM split planes, norms) for forwards that record no autograd
    graph: inference repeats the same ~40 small packing kernels per layer otherwise, which is most of a small graph's
    latency.  Keyed on every parameter's identity, storage and version, so any update rebuilds; one entry per tag."""
    params = [p for l in layers for p in l.parameters()]
    if layers[0].training or (torch.is_grad_enabled() and any(p.requires_grad for p in params)):
        return build()          # only eval-mode, graph-free forwards reuse (writes through `param.data` do not bump
                                # the version counter: call layers.clear_weight_cache(module) after such an edit)
    key = tuple((id(p), p.data_ptr(), p._version) for p in params)
    if os.environ.get("DISGAT_STRICT_CACHE") == "1":
        # writes through `param.data` do not bump the version counter: with the strict switch the key also carries a
        # checksum of the parameter VALUES (one small reduction + one host read per use - for drop-in users who edit
        # weights in place and cannot call clear_weight_cache)
        with torch.no_grad():
            key += (tuple(torch.stack([p.detach().double().sum() + p.detach().double().abs().sum() * 3.0 for p in params]).tolist()),)
    cache = layers[0].__dict__.setdefault("_disgat_memo", {})
    hit = cache.get(tag)
    if hit is not None and hit[0] == key:
        return hit[1]
    val = build()
    cache[tag] = (key, val)
    return val


def clear_weight_cache(module):
    """Drop the packed-weight caches of every layer under `module` (needed only after editing parameters through
    `.data`, which the cache key cannot see)."""
    for m in module.modules():
        m.__dict__.pop("_disgat_memo", None)


class _PackAtt3(torch.autograd.Function):
    """The H heads' att-3 parameters W_h [2 F_in, F_out], a_h [F_out, 1] as the operands of the fused pass:
    wt, wb [F_in, H*F_out] (top / bottom half of every W_h side by side) and a flat a_vec [H*F_out].
    Forward: the W_h are stacked along dim 1 - [2 F_in, H, F_out] IS [wt; wb], so both are row ranges of one buffer
    (one launch; the stack / slice / permute / reshape chain was 3).  Backward: the two operand gradients go back into
    one [H, 2 F_in, F_out] buffer whose slabs are the per-head gradients (autograd's own route through the slices
    zero-filled and copied a full-size tensor per slice: 7 launches per layer of a launch-bound step)."""

    @staticmethod
    def forward(ctx, f_in, *params):
        H = len(params) // 2
        wcat = torch.stack(params[:H], dim=1)                       # [2 F_in, H, F_out]
        f_out = wcat.shape[2]
        ctx.dims = (f_in, H, f_out)
        a_vec = torch.stack(params[H:]).reshape(-1)                 # [H, F_out, 1] -> flat
        return wcat[:f_in].view(f_in, H * f_out), wcat[f_in:].view(f_in, H * f_out), a_vec

    @staticmethod
    def backward(ctx, g_wt, g_wb, g_a):
        f_in, H, f_out = ctx.dims
        g_w = [None] * H
        if g_wt is not None or g_wb is not None:
            halves = [(g if g is not None else torch.zeros_like(o)).reshape(f_in, H, f_out).permute(1, 0, 2)
                      for g, o in ((g_wt, g_wb), (g_wb, g_wt))]
            g_w = list(torch.cat(halves, dim=1).unbind(0))          # [H, 2 F_in, F_out]: slab h is head h's gradient
        g_as = [None] * H if g_a is None else list(g_a.reshape(H, f_out, 1).unbind(0))
        return (None, *g_w, *g_as)


class _PackAtt1(torch.autograd.Function):
    """The H heads' att-1 parameters W_h [F_in, F_out], a_h [2 F_out, 1] as ONE operand of the score GEMM
    (layers.py:349-353: e = [h_r || h_c] . a = s1[r] + s2[c], s1 = x (W a_top), s2 = x (W a_bot)):
    wp [F_in, NP], columns [0, Hp) = W_h a_top,h, [Hp, 2 Hp) = W_h a_bot,h, the rest zero (NP = the GEMM kernels' column
    granule, so that x @ wp runs on them instead of a 2 H-wide library GEMM at the vector rate).  All heads at once: two
    stacks, one product + reduction - the per-head `W @ a[:F]` chain was 2 H matrix-vector launches (rocBLAS gemv) plus
    stacks per layer and pass, 3 H more in its backward."""

    @staticmethod
    def forward(ctx, Hp, NP, *params):
        H = len(params) // 2
        w = torch.stack(params[:H])                                  # [H, F_in, F_out]
        a2 = torch.stack(params[H:]).view(H, 2, 1, -1)               # [H, 2, 1, F_out]: top, bottom half of every a_h
        s = (w.unsqueeze(1) * a2).sum(-1)                            # [H, 2, F_in]
        wp = w.new_zeros((w.shape[1], NP))
        wp[:, :2 * Hp].view(-1, 2, Hp)[:, :, :H] = s.permute(2, 1, 0)
        ctx.save_for_backward(w, a2)
        ctx.dims = (H, Hp)
        return wp

    @staticmethod
    def backward(ctx, g):
        w, a2 = ctx.saved_tensors
        H, Hp = ctx.dims
        gs = g[:, :2 * Hp].reshape(-1, 2, Hp)[:, :, :H].permute(2, 1, 0)      # [H, 2, F_in]
        gw = (gs.unsqueeze(-1) * a2.view(H, 2, 1, -1)).sum(1)                  # [H, F_in, F_out] = sum_s gs[h,s,k] a[h,s,f]
        ga = (gs.unsqueeze(-1) * w.unsqueeze(1)).sum(2)                         # [H, 2, F_out]   = sum_k gs[h,s,k] W[h,k,f]
        return (None, None, *gw.unbind(0), *ga.reshape(H, -1, 1).unbind(0))


class _PackAtt2(torch.autograd.Function):
    """The H heads' att-2 score matrices W_h W_h^T (layers.py:362-365: e = <x_r W, x_c W> = <x_r (W W^T), x_c>) side by
    side, [F_in, Hp * F_in_p]: ONE head-batched product on the path's own GEMM (ops_gemm: the result lands in the
    concatenated layout), instead of H library GEMMs + pads + a cat per layer and pass.  Backward: grad W_h =
    (G_h + G_h^T) W_h, again one batched product."""

    @staticmethod
    def forward(ctx, Hp, f_in_p, *params):
        H = len(params)
        w = torch.stack(params)                                      # [H, F_in, F_out]
        f_in = w.shape[1]
        # (weights of the bundled graphs' size: one library bmm - their steps are launch-bound and the path's own GEMM takes
        # three preparation launches per operand)
        small = w.numel() <= 131072
        if small:
            m = torch.bmm(w, w.transpose(1, 2)).permute(1, 0, 2).reshape(f_in, H * f_in)
        else:
            m = ops_gemm._forward(w, w.transpose(1, 2), None, None, ops_gemm.ACT_NONE, 0.0)  # [F_in, H * F_in]
        if Hp != H or f_in_p != f_in:
            m = F.pad(m.view(f_in, H, f_in), (0, f_in_p - f_in, 0, Hp - H)).reshape(f_in, Hp * f_in_p)
        ctx.save_for_backward(w)
        ctx.dims = (H, Hp, f_in, f_in_p)
        return m

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        H, Hp, f_in, f_in_p = ctx.dims
        g3 = g.view(f_in, Hp, f_in_p)[:, :H, :f_in].permute(1, 0, 2)          # [H, F_in, F_in] view
        gs = g3 + g3.transpose(1, 2)
        if w.numel() <= 131072:
            return (None, None, *torch.bmm(gs, w).unbind(0))
        gw = ops_gemm._forward(gs, w, None, None, ops_gemm.ACT_NONE, 0.0)       # [F_in, H * F_out]
        return (None, None, *gw.view(f_in, H, -1).unbind(1))


def _pack_score_operands(layers, x, x_all, att, H, Hp, f_in, f_out, fp, am=None, cols=None):
    """Dense, differentiable (torch ops -> MFMA GEMMs) preparation of the per-node score operands.
    x: the rows this process owns; x_all: every node a column index can name (== x unsharded).

    att 1 (layers.py:349-353):  e = [h_r || h_c] . a,  h = x W   ==  s1[r] + s2[c]
                                with s1 = x (W a[:F]),  s2 = x (W a[F:])           -> [N,Hp] each
    att 2 (layers.py:362-365):  e = <x_r W, x_c W> = <x_r (W W^T), x_c>           -> P = x (W W^T)
    att 4 (= att 2 for inputs wider than the att-2 register tile): the same score from h = x W on both
                                sides, e = <h[r][h], h[c][h]>                      -> [N,Hp*fp], att-3 layout
    att 3 (layers.py:374-379):  e = a . lrelu([x_r || x_c] W) = a . lrelu(P[r] + Q[c]),
                                P = x W[:F_in], Q = x W[F_in:]                     -> [N,Hp*fp] each
    Returns (rowop, colop, a_vec); fp = padded per-head width of the att-3 operands.  am: max |x_all| as a device
    scalar (or None), the scale input of the f16x3 GEMMs (x's rows are a subset of x_all's, so it serves both).
    cols = (c0, c1) (att 3 / 4): only output features [c0, c1) of every head - a head wider than MAX_HEAD_SLICE is scored
    in feature slices whose partial scores add (the score is a sum over features).
    """
    c0, c1 = (0, f_out) if cols is None else cols
    fw = c1 - c0
    if att == 1:
        # one operand for both sides: s12 = x wp, s1 = columns [0, Hp), s2 = columns [Hp, 2 Hp) (strided views - the kernels
        # take row strides); NP = 32-column granule of the K <= 256 GEMM kernel
        NP = -(-2 * Hp // 32) * 32

        def pack1():
            wp = _PackAtt1.apply(Hp, NP, *[l.W for l in layers], *[l.a for l in layers])
            return wp, ops_gemm.presplit(wp)

        wp, sp = _memo(layers, ("att1", Hp), pack1)
        if x_all is x:
            s12 = ops_gemm.linear(x, wp, a_amax=am, w_split=sp)
            return s12[:, :Hp], s12[:, Hp:2 * Hp], None
        s_row = ops_gemm.linear(x, wp, a_amax=am, w_split=sp)                    # own rows: the gather may still be arriving
        s_col = ops_gemm.linear(parallel.finish(x_all), wp, a_amax=am, w_split=sp)
        return s_row[:, :Hp], s_col[:, Hp:2 * Hp], None
    if att == 4:
        # att 2 as the reference writes it (layers.py:362-365): h = x W per head, e = <h[r], h[c]>.  One operand table
        # serves both sides (column ids index the gathered x_all; unsharded, x_all is x and the GEMM runs once).
        def pack4():
            wc = torch.cat([F.pad(l.W[:, c0:c1], (0, fp - fw)) for l in layers] + [x.new_zeros(f_in, fp)] * (Hp - H), dim=1)
            return wc, ops_gemm.presplit(wc)

        wc, sc = _memo(layers, ("att4", fp, Hp, c0, c1), pack4)
        if x_all is x:
            hcol = ops_gemm.linear(x_all, wc, a_amax=am, w_split=sc)
            return hcol, hcol, None
        hrow = ops_gemm.linear(x, wc, a_amax=am, w_split=sc)                      # own rows first: the gather may still be arriving
        return hrow, _col_operand(x_all, wc, am, sc), None
    if att == 2:
        f_in_p = (f_in + 3) // 4 * 4

        def pack2():
            m = _PackAtt2.apply(Hp, f_in_p, *[l.W for l in layers])
            return m, ops_gemm.presplit(m)

        m, sm = _memo(layers, ("att2", Hp, f_in_p), pack2)
        return ops_gemm.linear(x, m, a_amax=am, w_split=sm), None, None             # [N, Hp*F_in_p]
    def pack3():
        if Hp == H and fp == fw == f_out:
            # no padding anywhere (the common case): one stack + two strided copies instead of 3 H pad / cat kernels -
            # on Cora-sized graphs the per-head form was a third of a train_step's launches
            wt, wb, a_vec = _PackAtt3.apply(f_in, *[l.W for l in layers], *[l.a for l in layers])
            return wt, wb, a_vec, ops_gemm.presplit(wt), ops_gemm.presplit(wb)
        tops = [F.pad(l.W[:f_in, c0:c1], (0, fp - fw)) for l in layers] + [x.new_zeros(f_in, fp)] * (Hp - H)
        bots = [F.pad(l.W[f_in:, c0:c1], (0, fp - fw)) for l in layers] + [x.new_zeros(f_in, fp)] * (Hp - H)
        a_vec = torch.cat([F.pad(l.a[c0:c1, 0], (0, fp - fw)) for l in layers] + [x.new_zeros(fp)] * (Hp - H))
        wt, wb = torch.cat(tops, dim=1), torch.cat(bots, dim=1)
        return wt, wb, a_vec.contiguous(), ops_gemm.presplit(wt), ops_gemm.presplit(wb)

    wt, wb, a_vec, st, sb = _memo(layers, ("att3", fp, Hp, c0, c1), pack3)
    p_row = ops_gemm.linear(x, wt, a_amax=am, w_split=st)         # own rows: runs while the all-gather's slices are on the links
    q_col = _col_operand(x_all, wb, am, sb)
    # how to rebuild either operand in the backward instead of keeping 2 x [N, H*F_out] floats alive until then (_remat_hooks)
    p_row._disgat_remat = (x, wt, am, st)
    q_col._disgat_remat = (x_all, wb, am, sb)
    return p_row, q_col, a_vec


# The att-3 score operands P = x W_top, Q = x_all W_bot ([N, H*F_out] each: 8 GB apiece at C4) are outputs of one GEMM
# and inputs of one fused pass.  Saved for the backward they were the largest thing a training step held (2 layers x 2 x
# 8 GB of a 119 GiB peak), and all the backward wanted from them was grad a = sum P (.) u + sum Q (.) v.  With
# REMAT_SCORE_OPERANDS (default) the fused pass owns the two GEMMs instead (ops.LayerPass + ops_bwd.layer_backward_u): the
# segment passes return u without the a-scale, G = x^T u is the weight-gradient GEMM that runs anyway, and
# grad W = G (.) a, grad a = sum_rows W (.) G, grad x = u (W (.) a)^T - no operand is saved, rebuilt or read.  The few passes
# that stay on ops.EdgePass (feature / column slices) rebuild their operands through saved-tensor hooks.  DISGAT_REMAT=0
# keeps the operands (ops_bwd.layer_backward).
REMAT_SCORE_OPERANDS = os.environ.get("DISGAT_REMAT", "1") != "0"


def _remat_hooks(*operands):
    table = {t.data_ptr(): (tuple(t.shape), t._disgat_remat) for t in operands
             if t is not None and getattr(t, "_disgat_remat", None) is not None}

    def pack(t):
        ent = table.get(t.data_ptr()) if (t.is_cuda and t.dim() == 2) else None
        if ent is not None and tuple(t.shape) == ent[0] and t.is_contiguous():
            return ("disgat-remat", ent[1])
        return t

    def unpack(o):
        if isinstance(o, tuple) and len(o) == 2 and o[0] == "disgat-remat":
            a, w, am, ws = o[1]
            with torch.no_grad():
                return ops_gemm._forward(a.detach(), w.detach(), None, None, ops_gemm.ACT_NONE, 0.0, am, ws)
        return o

    return torch.autograd.graph.saved_tensors_hooks(pack, unpack)


def _saved_operand_policy(rowop, colop):
    """Context for a fused pass that saves its score operands: rematerialise them in the backward when they carry a
    recipe (att 3, one head group) and the switch is on; otherwise a no-op context."""
    import contextlib
    if REMAT_SCORE_OPERANDS and torch.is_grad_enabled() and any(getattr(t, "_disgat_remat", None) is not None for t in (rowop, colop)):
        return _remat_hooks(rowop, colop)
    return contextlib.nullcontext()


def _col_operand(x_all, w, am, w_split):
    """x_all @ w for the column-side score operand; a gathered table still arriving (parallel.exchange(pipelined=True))
    is consumed slice by slice as it lands."""
    if parallel.pending_of(x_all) is not None:
        return parallel.project_gathered(x_all, w, am, w_split)
    return ops_gemm.linear(x_all, w, a_amax=am, w_split=w_split)


def _no_graph(layers, x):
    """True when this forward records no autograd graph (inference / torch.no_grad()): the plane-operand GEMM chain
    (ops_gemm.linear_planes, no backward) may replace the differentiable one."""
    return not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for l in layers for p in l.parameters())))


class StepSeed:
    """Attention-dropout seeds of a train_step that must replay from a HIP graph: the host part of a seed is fixed per
    call site (the k-th disga_heads call of the step), the per-step variation is a device counter the step advances itself
    (disgat_edge_fwd / disgat_bwd_alpha add it to the seed when they run).  The counter starts from torch's CPU generator,
    so torch.manual_seed still repeats a run."""
    GOLDEN = 0x9E3779B97F4A7C15 - (1 << 64)        # as a signed 64-bit addend

    def __init__(self, device):
        self.counter = torch.randint(0, 2 ** 62, (1,)).to(device)          # int64 storage, read as uint64 by the kernels
        self.site = 0

    def begin_step(self):
        self.counter.add_(self.GOLDEN)         # wraps in two's complement = the uint64 sum the kernels form
        self.site = 0

    def next_site(self):
        self.site += 1
        return (self.site * 0xD1B54A32D192ED03) & (2 ** 62 - 1)


STEP_SEED = None       # set by capture.StaticStep around a step's forward + backward


def disga_heads(layers, x, adj, aux_indices=None, head_ranges=None, aux_only=False, heads_planes=False, heads_discarded=False,
                heads_deferrable=False, defer_join=False):
    """All heads of one layer in fused passes: see _disga_heads (this wrapper only restores the autograd mode, which the
    implementation switches off for the tail of a layer whose per-head outputs the caller discards)."""
    prev = torch.is_grad_enabled()
    try:
        out = _disga_heads(layers, x, adj, aux_indices, head_ranges, aux_only, heads_planes, heads_discarded, heads_deferrable)
        if not defer_join:
            ops.join_side()        # (defer_join: the caller joins after its own GEMMs - DISGAT._run - so that they overlap the scorer too)
        return out
    finally:
        torch.set_grad_enabled(prev)


def _disga_heads(layers, x, adj, aux_indices=None, head_ranges=None, aux_only=False, heads_planes=False, heads_discarded=False,
                 heads_deferrable=False):
    """All H heads of one DISGAT layer: the loop of models.py:225-228 as ONE fused edge pass.

    layers: the H DisGALayer modules (parameter holders).  adj: torch sparse COO or CSRGraph.
    aux_indices: optional list of int64 (2,M_l) pair lists (layers.py:341).  head_ranges:
    optional list of (lo,hi) per aux list restricting which heads are scored on it (DisEdge uses
    half the heads per list, pretrainer.py:619-620); unscored heads' entries are None.
    aux_only: score the aux pairs only (no edge pass, no aggregation): returns (None, None, aux).
    heads_discarded: the caller computes this layer's aggregation / per-head outputs / edge scores only because the
    reference does and then drops them (predict_adjs_sparse's second layer, models.py:319-330): with autograd on, the
    fused layer pass then records no sign words for the edge list and everything after it runs without a graph - no
    gradient can reach those values, the aux scores keep theirs.
    heads_planes: the caller's consumer of the heads takes ops_gemm.Planes (our FuseLayer, the DifHead classifier): on
    a no-graph forward the head buffer is then written ONLY as planes (HeadList.planes; the list itself stays empty).
    heads_deferrable (with heads_planes): nothing but that consumer reads the heads - the projection is left to it
    (HeadList.deferred: a FuseLayer runs projection + ELU + its own GEMM as one launch, csrc/gemm_b2b.hip).
    Returns (HeadList of elu(h') [N,F_out], [edge_e[E,1]]*H, [[aux_e[M_l,1]]_l]*H or None).
    """
    grad_mode = torch.is_grad_enabled()
    l0 = layers[0]
    att, gnn = l0.att_type, l0.gnn_type
    H = len(layers)
    f_in, f_out = l0.in_features, l0.out_features
    if x.dim() != 2 or x.shape[1] != f_in:
        raise ValueError(f"expected a 2-D [N,{f_in}] input (layers.py:475 asserts 2-D), got {tuple(x.shape)}")
    if not x.is_cuda:
        raise RuntimeError("DISGAT HIP path: input must live on the GPU; there is no CPU fallback")
    drop = (0.0, 0)
    if l0.training and l0.dropout > 0:
        # attention dropout (layers.py:394) is generated inside the kernel from a counter-based hash;
        # the seed comes from torch's CPU generator, so torch.manual_seed makes runs repeatable.
        if STEP_SEED is not None:      # a step that may be captured in a HIP graph (capture.StaticStep): see StepSeed
            drop = (float(l0.dropout), STEP_SEED.next_site(), STEP_SEED.counter)
        else:
            drop = (float(l0.dropout), int(torch.randint(0, 2 ** 62, (1,)).item()))
    graph = graph_of(adj)
    if graph.n != x.shape[0]:
        raise ValueError("adjacency / feature row count mismatch")
    Hp = max(2, _pow2ceil(H))
    f_in_p = (f_in + 3) // 4 * 4
    if att == 2 and f_in_p > 512:
        # att 2's fused form dots P[r] = x_r (W W^T) with the gathered x[c] over all F_in columns held in registers
        # (F_in <= 512).  Wider inputs (raw bag-of-words features, --origin_feat: Cora 1 433) take the reference's
        # own formulation instead: per-head projections h = x W on both sides (kernel code 4), x aggregated in
        # column slices like att 1 / 3.
        att = 4
    # A head wider than one launch covers (att 3 / 4: 1024 features) is scored in equal feature slices: the score is a
    # sum over features, so the slices' partial scores add - all but the last slice run through the pair scorer on the
    # graph's own edge list, the last one through the edge pass with the partial sum as an additive input (e_in).
    f_slices = [(0, f_out)]
    if att in (3, 4) and f_out > MAX_HEAD_SLICE:
        n_sl = -(-f_out // MAX_HEAD_SLICE)
        fs = -(-f_out // n_sl)
        f_slices = [(k * fs, min(f_out, (k + 1) * fs)) for k in range(n_sl)]
    Hk, fp = _kernel_heads(att, Hp, f_slices[0][1] - f_slices[0][0], f_in_p)   # heads per launch; Hp / Hk head groups
    n_groups = Hp // Hk
    # register tile of the edge pass: Hk * ceil(F_in/256) float4 accumulators <= 16.  Wider inputs (raw
    # bag-of-words features, --origin_feat) are aggregated in column slices; the scores of att 1 / 3 do
    # not depend on x, so every slice sees identical attention weights (the scores are recomputed per
    # slice: correct, not cheap).  att 2's score is a dot product over all of x and cannot be sliced.
    tile = min(512, 256 * max(1, 16 // Hk))
    assert not (att == 2 and f_in_p > tile), "att 2 wider than its register tile is routed to the projected form above"
    # sharded: one exchange per layer - all rows, or only the referenced ones for edge-only passes
    x_all, graph = parallel.exchange(x, graph, edge_only=aux_indices is None, pipelined=True)

    # Scale inputs of the f16x3 GEMMs.  max |x| is measured once (1 pass over the layer input); the two 8x larger
    # operands get analytic upper bounds instead of their own pass: every Z row is a non-negative combination of x
    # rows with total weight <= 1/(1-p) (softmax weights, attention dropout rescaling; SageConv only divides further),
    # and a GEMM output is bounded by its input bound times the weight's largest column abs-sum.  A loose bound costs
    # no precision (the scheme is exact to 2^-23 per element down to 2^-27 of the scale), only range.
    if parallel.pending_of(x_all) is not None:      # the gathered table is still arriving: max over the ranks' own rows instead
        am_x = ops_gemm.amax_for(x)
        if am_x is not None:
            parallel.all_reduce_max(am_x, graph)
    else:
        am_x = ops_gemm.amax_for(x_all)
    z_bound = None if am_x is None else am_x * (1.001 / (1.0 - drop[0]))
    slice_ops = [_pack_score_operands(layers, x, x_all, att, H, Hp, f_in, f_out, fp, am_x,
                                      cols=None if len(f_slices) == 1 else sl) for sl in f_slices]
    rowop, colop, a_vec = slice_ops[-1]
    parallel.finish(x_all)                          # the edge pass / aux scorer gather rows of the whole table
    xg = x_all if (f_in_p == f_in and x_all.is_contiguous()) else F.pad(x_all, (0, f_in_p - f_in)).contiguous()
    # per-head operand width inside a row of rowop / colop (att 1: one scalar, att 2: F_in_p, att 3: fp)
    w_row = {1: 1, 2: f_in_p, 3: fp, 4: fp}[att]

    def group_ops(gi, si=-1):
        lo = gi * Hk * w_row
        hi = lo + Hk * w_row
        ro, co, avec = slice_ops[si]
        r = ro[:, lo:hi]
        c = None if co is None else co[:, lo:hi]
        av = None if avec is None else avec[lo:hi]
        d = (drop[0], drop[1] + 0x9E3779B1 * gi) if drop[0] > 0 else drop      # independent masks per group
        return r, c, av, d

    rec = ops.wants_sign(att, rowop, colop, a_vec)     # att 3: record lrelu signs for a gather-free backward

    def earlier_slices(gi, pairs, g_lo, g_hi):
        """Sum over all but the last feature slice of the partial scores of `pairs` (heads [g_lo, g_hi) of group gi);
        None for ordinary heads (one slice)."""
        tot = None
        for si in range(len(f_slices) - 1):
            r_, c_, av_, _d = group_ops(gi, si)
            part = ops.AuxPass.apply(None, r_, c_, av_, pairs, (att, Hk, f_in_p, fp, graph.n, g_lo, g_hi, rec))
            tot = part[g_lo:g_hi] if tot is None else tot + part[g_lo:g_hi]
        return tot

    if aux_indices is not None and not isinstance(aux_indices, list):
        aux_indices = [aux_indices]
    # differentiable att-3 layer with aux lists, one head group, no column slices (the training configuration): the edge
    # pass and all aux lists form ONE autograd node, so P / Q / a get one gradient each (ops.LayerPass)
    merged_aux = None
    recipes = (getattr(rowop, "_disgat_remat", None), getattr(colop, "_disgat_remat", None))
    own = bool(REMAT_SCORE_OPERANDS and att == 3 and None not in recipes)      # the pass can own the operands' GEMMs (ops.LayerPass)
    merge = rec and not aux_only and (aux_indices or own) and n_groups == 1 and f_in_p <= tile and len(f_slices) == 1
    heads = e_list = None
    concat = all(l.concat for l in layers)
    # plane-operand chain (csrc/gemm_planes.hip): edge pass -> Z planes -> projection -> head planes -> fuser.  One head
    # group, no column / feature slices, widths the kernel tiles (K = F_in a multiple of 32, 256 output columns per step)
    use_pl = (not aux_only and z_bound is not None and n_groups == 1 and len(f_slices) == 1 and f_in_p == f_in
              and f_in_p <= tile and concat and ops_gemm.planes_ok(f_in, f_out) and _no_graph(layers, x))
    def score_aux_lists():
        per_list = []
        for li, pairs in enumerate(aux_indices):
            lo, hi = (0, H) if head_ranges is None or head_ranges[li] is None else head_ranges[li]
            per_head = [None] * H
            if merged_aux is not None:
                for h in range(lo, hi):
                    per_head[h] = merged_aux[li][h].unsqueeze(1)
                per_list.append(per_head)
                continue
            for gi in range(n_groups):
                g_lo, g_hi = max(lo, gi * Hk), min(hi, (gi + 1) * Hk)      # heads of this group that are scored
                if g_lo >= g_hi:
                    continue
                r, c, av, _d = group_ops(gi)
                acfg = (att, Hk, f_in_p, fp, graph.n, g_lo - gi * Hk, g_hi - gi * Hk, rec)
                out = ops.AuxPass.apply(xg if att == 2 else None, r, c, av, pairs, acfg)
                if len(f_slices) > 1:          # wide heads: the other feature slices' partial scores
                    out = torch.cat([out[: g_lo - gi * Hk],
                                     out[g_lo - gi * Hk: g_hi - gi * Hk] + earlier_slices(gi, pairs, g_lo - gi * Hk, g_hi - gi * Hk),
                                     out[g_hi - gi * Hk:]])
                for h in range(g_lo, g_hi):
                    per_head[h] = out[h - gi * Hk].unsqueeze(1)
            per_list.append(per_head)
        return [[per_list[li][h] for li in range(len(aux_indices))] for h in range(H)]

    # No-graph forwards: the pair lists are scored NOW, on the CU-masked side stream (ops.run_on_side), beside the edge pass
    # and the dense GEMMs that follow it in this layer and the next - they need only the score operands packed above.
    early_aux = None
    if (aux_indices is not None and not aux_only and not merge and ops.overlap_enabled() and _no_graph(layers, x)
            and all(int(p.shape[1]) > 0 for p in aux_indices)):
        ops_in = [t for so in slice_ops for t in so if t is not None] + [xg] + list(aux_indices)
        early_aux = ops.run_on_side(score_aux_lists, ops_in)
        ops.mark_side_outputs([t for per_head in early_aux for t in per_head if t is not None])

    if not aux_only:
        z_groups, e_groups = [], []
        for gi in range(n_groups):
            r, c, av, d = group_ops(gi)
            if merge:
                lists = aux_indices or []
                ranges = tuple((0, H) if head_ranges is None or head_ranges[li] is None else tuple(head_ranges[li])
                               for li in range(len(lists)))
                cfg = (graph, att, Hk, f_in_p, fp, gnn == "SAGE", d, ranges)
                detach_edge = bool(heads_discarded and aux_indices)          # edge list: values only (see heads_discarded)
                if own:
                    # the pass owns P = x W_top and Q = x_all W_bot: their values go in detached, nothing of them is kept
                    # for the backward, which returns gradients for the GEMMs' inputs instead (ops_bwd.layer_backward_u)
                    (x_p, w_top, am_p, _s1), (x_q, w_bot, am_q, _s2) = recipes
                    z, edge_e, _den, *maux = ops.LayerPass.apply(xg, r.detach(), c.detach(), av, cfg + ((am_p, am_q), detach_edge),
                                                                 *lists, x_p, w_top, x_q, w_bot)
                else:
                    z, edge_e, _den, *maux = ops.LayerPass.apply(xg, r, c, av, cfg + (None, detach_edge), *lists)
                merged_aux = maux if aux_indices else None
                if detach_edge:
                    torch.set_grad_enabled(False)        # projection / fuser input of discarded heads: no graph (restored by the wrapper)
            elif use_pl:
                # no-graph forward: the aggregate leaves the edge pass as the two fp16 planes the projection GEMM consumes
                z, edge_e, _den = ops.edge_forward(graph, att, Hk, f_in_p, fp, xg, r, c, av, gnn == "SAGE", d, need_den=False,
                                                   z_bound=z_bound.reshape(1))
            elif f_in_p <= tile:
                cfg = (graph, att, Hk, f_in_p, fp, gnn == "SAGE", d, rec)
                e_in = earlier_slices(gi, graph.edge_pairs(), 0, Hk) if len(f_slices) > 1 else None
                with _saved_operand_policy(rowop, colop):
                    z, edge_e, _den = ops.EdgePass.apply(xg, r, c, av, cfg, e_in)
            else:
                zs, edge_e = [], None
                e_in = earlier_slices(gi, graph.edge_pairs(), 0, Hk) if len(f_slices) > 1 else None
                for c0 in range(0, f_in_p, tile):
                    c1 = min(f_in_p, c0 + tile)
                    cfg = (graph, att, Hk, c1 - c0, fp, gnn == "SAGE", d, rec)
                    zc, ec, _den = ops.EdgePass.apply(xg[:, c0:c1], r, c, av, cfg, e_in)
                    zs.append(zc)
                    edge_e = ec if edge_e is None else edge_e
                z = torch.cat(zs, dim=2)
            z_groups.append(z)
            e_groups.append(edge_e)
        z = z_groups[0] if n_groups == 1 else torch.cat(z_groups, dim=1)             # [N, Hp, F_in_p]
        edge_e = e_groups[0] if n_groups == 1 else torch.cat(e_groups, dim=0)        # [Hp, E]
        if merged_aux is not None or aux_indices is None or early_aux is not None:
            # nothing below scores anything any more: drop this frame's references to the score operands (2 x 8 GB at C4)
            # before the projection / fuser GEMMs allocate their outputs - with the rematerialising passes the operands
            # are then freed here, in inference at the latest
            slice_ops.clear()
            rowop = colop = r = c = None

        # ---- per-head output projection on the aggregated neighbourhood (dense, MFMA), written
        # straight into the fused [N, H*F_out] layout the fuser consumes (no torch.cat of heads)
        act_code = ops_gemm.ACT_ELU if concat else ops_gemm.ACT_NONE         # ELU fused in the GEMM epilogue
        if use_pl:
            zp = ops_gemm.Planes(z.hi[:, :H].permute(1, 0, 2), z.lo[:, :H].permute(1, 0, 2), z.bound)   # [H,N,F_in] views
            bias = init = None
            if gnn == "AT":                                                  # layers.py:397-399
                def pack_at_pl():
                    w = torch.stack([l.W_em for l in layers])
                    return ops_gemm.presplit_rm(w), w.detach().abs().sum(1).max()
                wr, wnorm = _memo(layers, "proj_AT_pl", pack_at_pl)
                pre_bound = z_bound * wnorm
            elif gnn == "SAGE":                                              # layers.py:96-110
                def pack_sage_pl():
                    wx = torch.cat([l.ag_layer.proj.weight[:, :f_in].t() for l in layers], dim=1)
                    wn = torch.stack([l.ag_layer.proj.weight[:, f_in:].t() for l in layers])
                    return wx.contiguous(), ops_gemm.presplit(wx), ops_gemm.presplit_rm(wn), wn.detach().abs().sum(1).max(), wx.detach().abs().sum(0).max()
                wx, wxs, wr, wn_norm, wx_norm = _memo(layers, "proj_SAGE_pl", pack_sage_pl)
                init = ops_gemm.linear(x, wx, a_amax=am_x, w_split=wxs)
                pre_bound = z_bound * wn_norm + am_x * wx_norm
            else:                                                            # layers.py:38-54
                def pack_gcn_pl():
                    w = torch.stack([l.ag_layer.weight for l in layers])
                    b = torch.cat([l.ag_layer.bias for l in layers]).contiguous()
                    return ops_gemm.presplit_rm(w), b, w.detach().abs().sum(1).max(), b.detach().abs().max()
                wr, bias, wnorm, bnorm = _memo(layers, "proj_GCN_pl", pack_gcn_pl)
                pre_bound = z_bound * wnorm + bnorm
            h_bound = torch.clamp(pre_bound * 1.001, min=1.0).reshape(1)     # |elu(v)| <= max(|v|, 1)
            if heads_planes and heads_deferrable and gnn in ("AT", "GCN") and ops_gemm.b2b_ok(f_in, f_out, 256):
                # only a FuseLayer will read these heads (DISGAT._run): leave the projection to it - with a fuser width the
                # back-to-back kernel takes, projection + ELU + fuser run as ONE launch and the head buffer never exists
                heads = HeadList()
                heads.deferred = DeferredHeads(zp, layers, gnn, wr, bias, h_bound, f_in, f_out)
                heads.fused_amax = h_bound
            else:
                act, hpl = ops_gemm.linear_planes(zp, wr, f_out, bias, init, act_code, 0.01, want_f32=not heads_planes, out_bound=h_bound)
                heads = HeadList() if act is None else HeadList(act[:, h * f_out:(h + 1) * f_out] for h in range(H))
                heads.fused, heads.planes, heads.fused_amax = act, hpl, h_bound
            heads.n_heads, heads.f_out = H, f_out
            e_list = [edge_e[h].unsqueeze(1) for h in range(H)]
            zt = None
        else:
            zt = z[:, :H, :f_in].permute(1, 0, 2)                           # [H,N,F_in] strided view
        fused_bound = None
        if use_pl:
            pass
        elif gnn == "AT":                                                    # layers.py:397-399
            def pack_at():
                w = torch.stack([l.W_em for l in layers])                    # [H,F_in,F_out]
                # the largest column abs-sum: small weights get it together with the output bound in one launch below
                return w, ops_gemm.presplit(w), (w.detach().abs().sum(1).max() if w.numel() > 131072 else None)

            w, ws, wnorm = _memo(layers, "proj_AT", pack_at)
            fused = ops_gemm.linear(zt, w, None, None, act_code, a_amax=z_bound, w_split=ws)
            pre_bound = None if z_bound is None or wnorm is None else z_bound * wnorm
            if z_bound is not None and wnorm is None:        # |elu(v)| <= max(|v|, 1), v = z W: bound z x column abs-sum
                fused_bound = ops_gemm.weight_bound(w, z_bound, 1.001, 1.0)[1]
        elif gnn == "SAGE":                                                  # layers.py:96-110
            wx = torch.cat([l.ag_layer.proj.weight[:, :f_in].t() for l in layers], dim=1)   # [F_in, H*F_out]
            wn = torch.stack([l.ag_layer.proj.weight[:, f_in:].t() for l in layers])        # [H,F_in,F_out]
            fused = ops_gemm.linear(zt, wn, None, ops_gemm.linear(x, wx, a_amax=am_x), act_code, a_amax=z_bound)
            pre_bound = None if z_bound is None else (z_bound * wn.detach().abs().sum(1).max()
                                                      + am_x * wx.detach().abs().sum(0).max())
        else:                                                                # layers.py:38-54
            w = torch.stack([l.ag_layer.weight for l in layers])
            b = torch.cat([l.ag_layer.bias for l in layers])                 # [H*F_out]
            fused = ops_gemm.linear(zt, w, b, None, act_code, a_amax=z_bound)
            pre_bound = None if z_bound is None else z_bound * w.detach().abs().sum(1).max() + b.detach().abs().max()
        if not use_pl:
            act = fused if concat else F.elu(fused)                          # layers.py:508-509
            heads = HeadList(act[:, h * f_out:(h + 1) * f_out] for h in range(H))
            heads.fused = act
            heads.n_heads, heads.f_out = H, f_out
            # |elu(v)| <= max(|v|, 1)
            heads.fused_amax = fused_bound if fused_bound is not None else (
                None if pre_bound is None else torch.clamp(pre_bound * 1.001, min=1.0).reshape(1))
            heads.pre_elu = None if concat else [fused[:, h * f_out:(h + 1) * f_out] for h in range(H)]
            e_list = [edge_e[h].unsqueeze(1) for h in range(H)]

    torch.set_grad_enabled(grad_mode)           # the aux scores below stay in the graph (heads_discarded only covers the tail above)
    aux_out = early_aux
    if aux_indices is not None and early_aux is None:
        aux_out = score_aux_lists()
    return heads, e_list, aux_out


class FuseLayer(nn.Module):
    """layers.py:876-921: cat heads (+residue) -> Linear -> leaky_relu, three residue_type variants."""

    def __init__(self, args, nheads, nfeat=64, residue=0):
        super().__init__()
        self.args = args
        self.nheads = nheads
        self.nfeat = nfeat
        self.residue_dim = residue
        if self.args.residue_type == 0:
            self.fuse = nn.Linear(self.nfeat * nheads + self.residue_dim, self.nfeat)
        if self.args.residue_type == 1:
            self.fuse = nn.Linear(self.nfeat * nheads + self.residue_dim, self.nfeat * 2)
            self.fuse2 = nn.Linear(self.nfeat * 2, self.nfeat)
        if self.args.residue_type == 2:
            self.fuse = nn.Linear(self.nfeat * nheads, self.nfeat)
            if self.residue_dim != 0:
                self.fuse2 = nn.Linear(self.residue_dim, self.nfeat)

    def accepts_planes(self):
        """True when forward() consumes HeadList.planes directly (plain Linear on the concatenated heads, no residue
        columns, a width the plane GEMM tiles): DISGAT._run then asks disga_heads not to write the fp32 head buffer."""
        return (self.args.residue_type == 0 and self.residue_dim == 0 and self.fuse.weight.is_cuda
                and ops_gemm.planes_ok(self.fuse.in_features, self.fuse.out_features))

    def forward(self, feature_list, residue=None):
        deferred = getattr(feature_list, "deferred", None)
        if deferred is not None:
            n2 = self.fuse.out_features
            if (self.accepts_planes() and not torch.is_grad_enabled() and self.fuse.in_features == deferred.H * deferred.f_out
                    and ops_gemm.b2b_ok(deferred.f_in, deferred.f_out, n2)):
                # projection -> ELU -> this Linear (+ leaky_relu) back to back: one launch, no head buffer
                act = ops_gemm.ACT_NONE if self.args.fuse_no_relu else ops_gemm.ACT_LEAKY
                # the chunk images hold BOTH weights: memoised on the fuser and the layer's heads together
                wch = _memo([self] + list(deferred.layers), ("b2b", deferred.gnn),
                            lambda: ops_gemm.presplit_b2b(deferred.w1_stack(), self.fuse.weight.t()))
                return ops_gemm.proj_fuse(deferred.zp, wch, deferred.bias, self.fuse.bias, deferred.h_bound, deferred.f_out, n2, act, 0.01)
            deferred.materialize(feature_list)
        planes = getattr(feature_list, "planes", None)
        if planes is not None and self.accepts_planes() and not torch.is_grad_enabled():
            act = ops_gemm.ACT_NONE if self.args.fuse_no_relu else ops_gemm.ACT_LEAKY
            ws = _memo([self], "fuse_rm", lambda: ops_gemm.presplit_rm(self.fuse.weight.t()))
            return ops_gemm.linear_planes(planes, ws, self.fuse.out_features, self.fuse.bias, None, act, 0.01)[0]
        features = getattr(feature_list, "fused", None)
        if features is None and planes is not None:
            features = planes.to_f32()
        if features is None:
            features = torch.cat(list(feature_list), dim=-1)
        use_res = self.residue_dim != 0 and residue is not None
        if self.args.residue_type == 0:
            if use_res:
                features = torch.cat([features, residue], dim=-1)
            if features.is_cuda and features.dim() == 2:     # Linear (+ leaky_relu) in one MFMA GEMM with fused epilogue
                act = ops_gemm.ACT_NONE if self.args.fuse_no_relu else ops_gemm.ACT_LEAKY
                am = None if use_res else getattr(feature_list, "fused_amax", None)
                ws = _memo([self], "fuse", lambda: ops_gemm.presplit(self.fuse.weight.t()))
                return ops_gemm.linear(features, self.fuse.weight.t(), self.fuse.bias, None, act, 0.01, a_amax=am, w_split=ws)
            feature = self.fuse(features)
        elif self.args.residue_type == 1:
            if use_res:
                features = torch.cat([features, residue], dim=-1)
            feature = self.fuse2(F.leaky_relu(self.fuse(features)))
        else:
            feature = self.fuse(features)
            if use_res:
                feature = feature + self.fuse2(residue)
        if not self.args.fuse_no_relu:
            feature = F.leaky_relu(feature)
        return feature
