"""Pinning the side taken at the leaky-ReLU kink of the att-3 score (gradient tests only).

e = a . leaky_relu(z), z = P[r] + Q[c].  Where |z| is within rounding of 0, the fp32 kernels and the float64 oracle
may evaluate z on different sides; the VALUE is unaffected (|z| * 0.99 < 1e-5) but that term's derivative flips
between 0.01 and 1 - a property of the function, not of either implementation.  With H * F_out * (E + M) ~ 1e6..1e9
arguments a few always sit that close, so a gradient comparison has to agree on the side first.

`record_operands()` wraps layers._pack_score_operands while the product runs and keeps the P / Q it computed (the
product's own GEMM output, fp32); `Pins` turns them into the sign of the fp32 sum for every (pair, head, feature) - the
very comparison the kernel makes - and `hook()` is installed as oracle.LRELU3: the oracle keeps its own float64 side
wherever |z64| >= thr and follows the recorded fp32 side below it.  Forward comparisons never need any of this.
"""
import contextlib

import torch


@contextlib.contextmanager
def record_operands(store):
    """Within the block every call of layers._pack_score_operands appends (rowop, colop) to `store`."""
    from edgedisentangle_ssl_amd import layers
    orig = layers._pack_score_operands

    def wrapped(*a, **k):
        out = orig(*a, **k)
        cols = k.get("cols")
        entry = (out[0].detach(), out[1].detach() if out[1] is not None else None, cols)
        if cols is None or cols[0] == 0:
            store.append([entry])               # a new layer (one entry per feature slice of its heads)
        else:
            store[-1].append(entry)
        return out
    layers._pack_score_operands = wrapped
    try:
        yield store
    finally:
        layers._pack_score_operands = orig


class Pins:
    """operands: per layer (in call order) the list of its feature slices' (rowop [N, Hp*fp], colop [N_all, Hp*fp],
    (c0, c1) or None) as record_operands() stores them; lists: the (rows, cols) index lists the oracle scores per
    head, in its call order (edges first, then every aux list)."""

    def __init__(self, operands, n_heads, f_out, lists, thr=1e-5):
        self.ops, self.H, self.f_out, self.lists, self.thr = operands, n_heads, f_out, lists, thr
        self.calls = 0
        self.pinned = 0
        self.disagree_far = 0

    def hook(self, z, negative_slope=0.01):
        k = self.calls
        self.calls += 1
        per_layer = self.H * len(self.lists)
        layer, rem = divmod(k, per_layer)
        h, li = divmod(rem, len(self.lists))
        hp = max(2, 1 << (self.H - 1).bit_length())
        r, c = self.lists[li]
        parts = []
        for rowop, colop, cols in self.ops[layer]:          # one entry, or one per feature slice of a wide head
            fp = rowop.shape[1] // hp
            width = self.f_out if cols is None else cols[1] - cols[0]
            dev = rowop.device
            sl = slice(h * fp, h * fp + width)
            parts.append(((rowop[:, sl][r.to(dev)] + colop[:, sl][c.to(dev)]) > 0).cpu())
        pos32 = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        zd = z.detach()
        near = zd.abs() < self.thr
        self.pinned += int(near.sum())
        self.disagree_far += int(((zd > 0) != pos32)[~near].sum())
        side = torch.where(near, pos32, zd > 0)
        slope = torch.where(side, torch.ones((), dtype=z.dtype), torch.full((), negative_slope, dtype=z.dtype))
        return z * slope


@contextlib.contextmanager
def pinned_oracle(pins):
    from oracle import disgat_oracle as orc
    orig = orc.LRELU3
    orc.LRELU3 = pins.hook if pins is not None else orig
    try:
        yield
    finally:
        orc.LRELU3 = orig
