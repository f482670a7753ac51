"""Dense contractions of the DISGAT path on the split-precision MFMA GEMMs (csrc/gemm_split.hip).

`linear(a, w, ...)` computes act(a @ w + bias + init) for a 2-D `a`, or for a head-batched view
`a` [H, M, K] with `w` [H, K, N] writing the concatenated-heads layout [M, H*N] (`init` is then
[M, H*N], or [M, N] shared by every head).  Shapes outside
the kernel's tiling (N % 128, K % 32) and DISGAT_GEMM=blas go to hipBLASLt through torch.matmul -
still the GPU, still fp32.  DISGAT_GEMM selects the scheme: `f16x3` (default: two fp16 planes per operand, 3
products, operands accurate to 2^-23 per element), `split6` (three bf16 planes, 6 products, exact operands),
`split3` (bf16, 3 products: not fp32-accurate, benchmark switch only), `blas`.
"""
import os

import torch
import torch.nn.functional as F

from . import _lib, ops

ACT_NONE, ACT_ELU, ACT_LEAKY = 0, 1, 2


def mode():
    return os.environ.get("DISGAT_GEMM", "f16x3")


def amax(a):
    """Device scalar max |a| of a 2-D or head-batched 3-D operand (input of the f16x3 scheme's scale)."""
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    if a.dim() == 3:
        hb, m, k = a.shape
        _lib.call("disgat_amax", a.data_ptr(), a.stride(1), a.stride(0), m, k, hb, out.data_ptr(), ops._stream())
    else:
        m, k = a.shape
        _lib.call("disgat_amax", a.data_ptr(), a.stride(0), 0, m, k, 1, out.data_ptr(), ops._stream())
    return out


def amax_for(a):
    """amax(a) to share between several linear() calls on the same operand (or on row subsets of it: any upper
    bound of max |a| is a valid scale input); None when the f16x3 kernel would not be used."""
    if mode() != "f16x3" or not _kernel_ok(a, a.shape[-1], 128):
        return None
    return amax(a)


def split_weight_f16(w):
    """[..., K, N] fp32 -> (int16 view of fp16 planes [..., 2, N, K] = hi, lo of w^T * s, device scalar s).
    s is the power of two that puts max |w| in [2^13, 2^14); lo carries an extra 2^11 (see gemm_f16x3_kernel)."""
    w = w.detach()
    if w.dim() == 2:
        w = w.unsqueeze(0)
    hb, k, n = w.shape
    planes = torch.empty((hb, 2, n, k), dtype=torch.int16, device=w.device)
    amax_scale = torch.empty(2, dtype=torch.float32, device=w.device)
    _lib.call("disgat_split_f16", w.data_ptr(), w.stride(0), w.stride(1), w.stride(2), k, n, hb, planes.data_ptr(),
              amax_scale.data_ptr(), ops._stream())
    return planes, amax_scale[1:]


def split_weight(w):
    """[..., K, N] fp32 -> int16 view of bf16 planes [..., 3, N, K] (hi, mid, lo of w^T, k contiguous)."""
    wt = w.detach().transpose(-1, -2).contiguous()
    hi = wt.to(torch.bfloat16)
    r1 = wt - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return torch.stack([hi, mid, lo], dim=-3).contiguous().view(torch.int16)


def weight_bound(w, in_bound=None, scale=1.0, floor=0.0):
    """(largest column abs-sum of w [K,N] / [H,K,N] as a device scalar, max(floor, in_bound * that * scale) or None): the scale
    bound of x @ w given a bound on |x| - one launch for small weights (disgat_weight_bound), the ATen chain otherwise."""
    w = w.detach()
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    if w3.is_cuda and w3.dtype == torch.float32 and w3.numel() <= 131072 and (in_bound is None or in_bound.dtype == torch.float32):
        out = torch.empty(2, dtype=torch.float32, device=w.device)
        hb, k, n = w3.shape
        _lib.call("disgat_weight_bound", w3.data_ptr(), w3.stride(0), w3.stride(1), w3.stride(2), k, n, hb, ops._ptr(in_bound),
                  float(scale), float(floor), out.data_ptr(), ops._stream())
        return out[0], (out[1:2] if in_bound is not None else None)
    norm = w3.abs().sum(1).max()
    return norm, (None if in_bound is None else torch.clamp(in_bound * norm * scale, min=floor).reshape(1))


def _apply_act(t, act, slope):
    if act == ACT_ELU:
        return F.elu(t)
    if act == ACT_LEAKY:
        return F.leaky_relu(t, slope)
    return t


def _tiles(k, n):
    """Output / reduction widths the f16x3 launcher takes: N % 128 (K % 32), or N % 32 for K = 64 / 128 / 256 (the
    register-stationary kernel, csrc/gemm_rs.hip - what lets the nhid = 64 layers of the bundled graphs run on it)."""
    return k % 32 == 0 and (n % 128 == 0 or (mode() == "f16x3" and k in (64, 128, 256) and n % 32 == 0 and n >= 32))


def _kernel_ok(a, k, n):
    return (mode() != "blas" and a.is_cuda and a.dtype == torch.float32 and _tiles(k, n)
            and a.stride(-1) == 1 and a.stride(-2) % 4 == 0 and a.data_ptr() % 16 == 0
            and (a.dim() == 2 or a.stride(0) % 4 == 0))


def presplit(w):
    """Split planes of a weight for reuse across calls (inference: the weights do not change between forwards);
    None when the active scheme has nothing to precompute for it."""
    if mode() != "f16x3" or not w.is_cuda or not _tiles(w.shape[-2], w.shape[-1]):
        return None
    return split_weight_f16(w)


class Planes:
    """An fp32 operand stored ALREADY SPLIT for the f16x3 GEMMs: `hi`, `lo` int16 views of fp16 planes with the logical
    tensor's shape and strides (2-D [M,K], or a head-batched [H,M,K] view of an [M,H,K] buffer), `bound` a device
    scalar >= max |v| - producer and consumer derive the common power-of-two scale from it (csrc/gemm_common.h)."""
    __slots__ = ("hi", "lo", "bound")

    def __init__(self, hi, lo, bound):
        self.hi, self.lo, self.bound = hi, lo, bound

    @property
    def shape(self):
        return self.hi.shape

    def dim(self):
        return self.hi.dim()

    def view_heads(self, nh):
        """[M, nh*K] planes as the head-batched [nh, M, K] view."""
        m, hk = self.hi.shape
        return Planes(self.hi.view(m, nh, hk // nh).permute(1, 0, 2), self.lo.view(m, nh, hk // nh).permute(1, 0, 2), self.bound)

    def to_f32(self):
        """(hi + lo 2^-11) / s as a fresh fp32 tensor (every element to 2^-23 of itself)."""
        h = self.hi
        b = h.dim() == 3
        hb, m, k = (h.shape if b else (1,) + tuple(h.shape))
        out = torch.empty((m, hb, k) if b else (m, k), dtype=torch.float32, device=h.device)
        ov = out.permute(1, 0, 2) if b else out
        _lib.call("disgat_planes_to_f32", h.data_ptr(), self.lo.data_ptr(), h.stride(-2), h.stride(0) if b else 0, m, k, hb,
                  self.bound.data_ptr(), ov.data_ptr(), ov.stride(-2), ov.stride(0) if b else 0, ops._stream())
        return ov if b else out


def planes_ok(k, n):
    """Shapes disgat_gemm_planes takes (its tiling: 256 output columns per step, 32-deep k-steps).  DISGAT_PLANES=0
    switches the plane-operand chain off (same-box A/B against the fp32-operand kernels)."""
    return mode() == "f16x3" and os.environ.get("DISGAT_PLANES", "1") != "0" and n % 256 == 0 and k % 32 == 0 and k >= 64


def split_planes(x, bound=None):
    """fp32 [M,K] or head-batched [H,M,K] view -> Planes with the same logical layout (contiguous [M,K] / [M,H,K])."""
    b = x.dim() == 3
    hb, m, k = (x.shape if b else (1,) + tuple(x.shape))
    if bound is None:
        bound = amax(x)
    hi = torch.empty((m, hb, k) if b else (m, k), dtype=torch.int16, device=x.device)
    lo = torch.empty_like(hi)
    hv, lv = (hi.permute(1, 0, 2), lo.permute(1, 0, 2)) if b else (hi, lo)
    _lib.call("disgat_split_planes", x.data_ptr(), x.stride(-2), x.stride(0) if b else 0, m, k, hb, bound.data_ptr(),
              hv.data_ptr(), lv.data_ptr(), hv.stride(-2), hv.stride(0) if b else 0, ops._stream())
    return Planes(hv, lv, bound)


def presplit_rm(w):
    """Row-major [.., 2, N, K] weight planes + scale for disgat_gemm_planes (any K); None when it cannot take the shape."""
    if not w.is_cuda or not planes_ok(w.shape[-2], w.shape[-1]):
        return None
    w = w.detach()
    if w.dim() == 2:
        w = w.unsqueeze(0)
    hb, k, n = w.shape
    planes = torch.empty((hb, 2, n, k), dtype=torch.int16, device=w.device)
    amax_scale = torch.empty(2, dtype=torch.float32, device=w.device)
    _lib.call("disgat_split_f16_rm", w.data_ptr(), w.stride(0), w.stride(1), w.stride(2), k, n, hb, planes.data_ptr(),
              amax_scale.data_ptr(), ops._stream())
    return planes, amax_scale[1:]


def b2b_ok(k1, n1, n2):
    """Shapes disgat_proj_fuse takes (csrc/gemm_b2b.hip): per-head projection [k1 -> n1] + fuser [H*n1 -> n2] in one launch.
    DISGAT_B2B=0 keeps the two-launch plane chain (same-box A/B)."""
    return (mode() == "f16x3" and os.environ.get("DISGAT_B2B", "1") != "0" and k1 in (64, 128, 256)
            and n1 % 32 == 0 and 64 <= n1 <= 256 and n2 in (64, 128, 256))


_B2B_PERM = {}


def presplit_b2b(w1, w_fuse_t):
    """Weights of disgat_proj_fuse as ONE stream of chunk images (csrc/gemm_b2b.hip): w1 [nh, K1, N1] stacked per-head
    projection weights, w_fuse_t = fuse.weight.t() [nh*N1, N2].  Both are split like any K <= 256 weight (fragment-major
    planes); the fuser weight first gets the rows of every group of 32 reordered to the order a lane of GEMM 1's accumulator
    tiles holds its columns in (position 8 q + e takes row 4 q + e for e < 4, row 16 + 4 q + e - 4 for e >= 4).  Chunk (h, j)
    = [W1 planes x n-tiles 2j, 2j+1 x K1/32 blocks | W2 planes x N2/16 blocks of k-step j], each block 1 KB, contiguous:
    the image a ring slot receives.  Returns (int16 [nh, N1/32, slot halfs], s1, s2)."""
    w1 = w1.detach()
    w2 = w_fuse_t.detach()
    nh, k1, n1 = w1.shape
    k2, n2 = w2.shape
    key = (k2, w2.device)
    idx = _B2B_PERM.get(key)
    if idx is None:
        kp = torch.arange(32)
        q, e = kp >> 3, kp & 7
        c = torch.where(e < 4, 4 * q + e, 16 + 4 * q + e - 4)
        idx = (torch.arange(k2 // 32)[:, None] * 32 + c[None, :]).reshape(-1).to(w2.device)
        _B2B_PERM[key] = idx
    p1, s1 = split_weight_f16(w1)                                            # [nh, 2, N1, K1]: blocks (n / 16, k / 32)
    p2, s2 = split_weight_f16(w2.index_select(0, idx).view(nh, n1, n2))      # [nh, 2, N2, N1]: blocks (n / 16, k / 32)
    nj, kt, nt2 = n1 // 32, k1 // 32, n2 // 16
    c1 = p1.view(nh, 2, nj, 2, kt, 512).permute(0, 2, 1, 3, 4, 5).reshape(nh, nj, 2 * 2 * kt * 512)
    c2 = p2.view(nh, 2, nt2, nj, 512).permute(0, 3, 1, 2, 4).reshape(nh, nj, 2 * nt2 * 512)
    return torch.cat([c1, c2], dim=2).contiguous(), s1, s2


def proj_fuse(zp, w_chunks, bias1, bias2, mid_bound, n1, n2, act2=ACT_LEAKY, slope=0.01):
    """act2(cat_h[elu(Z_h W1_h + bias1_h)] W2 + bias2) in one launch (no autograd).  zp: Planes [H, M, K1] view of the edge
    pass's aggregate; w_chunks = presplit_b2b(W1 stack, fuse.weight.t()); mid_bound: device scalar >= max |elu(.)|."""
    hb, m, k1 = zp.shape
    chunks, s1, s2 = w_chunks
    out = torch.empty((m, n2), dtype=torch.float32, device=zp.hi.device)
    if bias1 is not None:
        bias1 = bias1.contiguous()
    if bias2 is not None:
        bias2 = bias2.contiguous()
    ops._launch("disgat_proj_fuse", "gemm_b2b", 2.0 * m * hb * n1 * (k1 + n2),
                zp.hi.data_ptr(), zp.lo.data_ptr(), zp.hi.stride(-2), zp.hi.stride(0), zp.bound.data_ptr(),
                chunks.data_ptr(), s1.data_ptr(), ops._ptr(bias1), s2.data_ptr(), ops._ptr(bias2),
                mid_bound.data_ptr(), out.data_ptr(), out.stride(0), m, hb, k1, n1, n2, act2, float(slope), ops._stream())
    return out


def linear_planes(ap, w_rm, n, bias=None, init=None, act=ACT_NONE, slope=0.01, want_f32=True, out_bound=None):
    """act(A @ W + bias + init) with A given as Planes (no autograd: inference forwards only).  w_rm = presplit_rm(W).
    Returns (fp32 [M, H*N] or None, Planes [M, H*N] or None): the plane output (scaled by `out_bound`, a device scalar
    bounding |result|) is what the next GEMM of the chain consumes."""
    batched = ap.dim() == 3
    hb, m, k = (ap.shape if batched else (1,) + tuple(ap.shape))
    planes, b_scale = w_rm
    dev = ap.hi.device
    out = torch.empty((m, hb * n), dtype=torch.float32, device=dev) if want_f32 else None
    oh = ol = None
    if out_bound is not None:
        oh = torch.empty((m, hb * n), dtype=torch.int16, device=dev)
        ol = torch.empty_like(oh)
    init_bs = n if batched else 0
    if init is not None and batched and init.shape == (m, n) and init.stride(-1) == 1:
        init_bs = 0
    elif init is not None and (init.stride(-1) != 1 or tuple(init.shape) != (m, hb * n)):
        init = init.expand(m, hb * n).contiguous()
    if bias is not None:
        bias = bias.contiguous()
    ops._launch("disgat_gemm_planes", "gemm_f16x3", 2.0 * m * n * k * hb,
                ap.hi.data_ptr(), ap.lo.data_ptr(), ap.hi.stride(-2), ap.hi.stride(0) if batched else 0, planes.data_ptr(),
                ap.bound.data_ptr(), b_scale.data_ptr(), ops._ptr(bias), ops._ptr(init),
                0 if init is None else init.stride(0), init_bs, ops._ptr(out), hb * n, n if batched else 0,
                ops._ptr(oh), ops._ptr(ol), hb * n, n if batched else 0, ops._ptr(out_bound), m, n, k, hb, act, float(slope),
                ops._stream())
    return out, (None if oh is None else Planes(oh, ol, out_bound))


def logits_ok(k, n_hidden, n_out):
    """Shapes disgat_gemm_planes_logits takes (DISGAT_LOGITS=0: the two-launch form, same-box A/B)."""
    return planes_ok(k, n_hidden) and n_hidden == 256 and 1 <= n_out <= 16 and os.environ.get("DISGAT_LOGITS", "1") != "0"


def presplit_logits(w2, b2):
    """nn.Linear(256 -> n_out) weight [n_out, 256] (+ bias) as the MFMA fragment image disgat_gemm_planes_logits reads:
    (int16 [8, 2, 64, 8], device scalar s_W2, fp32 [16] bias or None, n_out).  Pure tensor ops, no host read."""
    w2 = w2.detach()
    n_out, k = w2.shape
    wt = torch.zeros((16, k), dtype=torch.float32, device=w2.device)
    wt[:n_out] = w2
    amax_w = wt.abs().max()
    _, e = torch.frexp(amax_w)                                             # gemm_common.h: f16_scale()
    s = torch.where((amax_w > 0) & torch.isfinite(amax_w), torch.ldexp(torch.ones_like(amax_w), 14 - e.clamp(-100, 100)),
                    torch.ones_like(amax_w))
    t = wt * s
    hi = t.to(torch.float16)
    lo = ((t - hi.float()) * 2048.0).to(torch.float16)
    # [p][o][k] -> [g = k / 32][p][lane = 16 (k % 32 / 8) + o][e = k % 8]
    img = torch.stack([hi, lo]).view(2, 16, k // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous().view(k // 32, 2, 64, 8)
    bias = None
    if b2 is not None:
        bias = torch.zeros(16, dtype=torch.float32, device=w2.device)
        bias[:n_out] = b2.detach()
    return img.view(torch.int16), s.reshape(1), bias, n_out


def linear_planes_logits(ap, w_rm, bias, init, act, slope, mid_bound, w2_prep):
    """act(A @ W1 + bias + init) @ W2^T + b2 in one launch, A given as head-batched Planes [H, M, K]; returns fp32
    [M * H, n_out], row (m, h) at m * H + h.  w_rm = presplit_rm(W1 [H, K, 256]), w2_prep = presplit_logits(...)."""
    batched = ap.dim() == 3
    hb, m, k = (ap.shape if batched else (1,) + tuple(ap.shape))
    planes, b_scale = w_rm
    img, s2, bias2, n_out = w2_prep
    n = 256
    out = torch.empty((m * hb, n_out), dtype=torch.float32, device=ap.hi.device)
    init_bs = n if batched else 0
    if init is not None and batched and init.shape == (m, n) and init.stride(-1) == 1:
        init_bs = 0
    elif init is not None and (init.stride(-1) != 1 or tuple(init.shape) != (m, hb * n)):
        init = init.expand(m, hb * n).contiguous()
    if bias is not None:
        bias = bias.contiguous()
    ops._launch("disgat_gemm_planes_logits", "gemm_f16x3", 2.0 * m * hb * (n * k + n * 16),
                ap.hi.data_ptr(), ap.lo.data_ptr(), ap.hi.stride(-2), ap.hi.stride(0) if batched else 0, planes.data_ptr(),
                ap.bound.data_ptr(), b_scale.data_ptr(), ops._ptr(bias), ops._ptr(init),
                0 if init is None else init.stride(0), init_bs, mid_bound.data_ptr(), img.data_ptr(), s2.data_ptr(),
                ops._ptr(bias2), out.data_ptr(), m, n, k, hb, n_out, act, float(slope), ops._stream())
    return out


_KPAD = {}          # (data_ptr, version, shape) -> (tensor kept alive, zero-padded copy): padded copies of constant operands


def _pad_to_tiles(a, w, k, n):
    """A large 2-D product whose K is not a multiple of 32 or whose N is off the kernels' column granule (raw bag-of-words
    features: Cora 1 433, cora_full 8 710 columns; /root/reference/main.py:108 takes the width from the data file) as a
    product the f16x3 kernels tile: K zero-padded to 32 - the padded copy of an operand that needs no gradient (the feature
    matrix: a run constant) is kept and reused while the tensor is unchanged - and the weight's columns zero-padded to the
    granule.  hipBLASLt's fp32 GEMM, where such shapes went, runs at the vector rate on CDNA4.  Returns (a', w', Kp, Np) or
    None when the shape is small (launch-bound: the library's one launch is the cheaper route) or already tiles."""
    if mode() != "f16x3" or a.dim() != 2 or not a.is_cuda or a.dtype != torch.float32 or os.environ.get("DISGAT_PAD_SHAPES", "1") == "0":
        return None
    m = a.shape[0]
    kp = -(-k // 32) * 32
    gran = 32 if kp in (64, 128, 256) else 128
    np_ = -(-n // gran) * gran
    if (kp == k and np_ == n) or m < 8192 or 2.0 * m * kp * np_ < 1e9 or np_ > 4 * n:
        return None
    if kp != k or a.stride(1) != 1 or a.stride(0) % 4 or a.data_ptr() % 16:
        key = (a.data_ptr(), a._version, tuple(a.shape), tuple(a.stride()))
        hit = _KPAD.get(key) if not a.requires_grad else None
        if hit is None:
            ap = F.pad(a.detach(), (0, kp - k))
            if not a.requires_grad:
                if len(_KPAD) >= 4:
                    _KPAD.pop(next(iter(_KPAD)))
                _KPAD[key] = (a, ap)
        else:
            ap = hit[1]
    else:
        ap = a
    wp = F.pad(w.detach(), (0, np_ - n, 0, kp - k)) if (kp != k or np_ != n) else w
    return ap, wp, kp, np_


def _forward(a, w, bias, init, act, slope, a_amax=None, w_split=None, out=None):
    """out: optional preallocated 2-D result (row-contiguous view, e.g. a row block of a larger table) to write into."""
    batched = a.dim() == 3
    if batched:
        hb, m, k = a.shape
        n = w.shape[2]
    else:
        hb, (m, k), n = 1, a.shape, w.shape[1]
        if out is None and not _kernel_ok(a, k, n):
            padded = _pad_to_tiles(a, w, k, n)
            if padded is not None:
                ap, wp, kp, np_ = padded
                bp = None if bias is None else F.pad(bias, (0, np_ - n))
                ip = None if init is None else F.pad(init.expand(m, n), (0, np_ - n))
                res = _forward(ap, wp, bp, ip, act, slope, a_amax if kp == k else None, None)
                return res if np_ == n else res[:, :n].contiguous()
    out_arg = out
    if out_arg is not None and (tuple(out_arg.shape) != (m, hb * n) or out_arg.stride(-1) != 1 or out_arg.dtype != torch.float32):
        raise RuntimeError("ops_gemm: `out` must be a float32 [M, N] view with unit inner stride")
    if not _kernel_ok(a, k, n):
        if batched:
            out3 = torch.bmm(a, w)
            if bias is not None:
                out3 = out3 + bias.view(hb, 1, n)
            out = out3.permute(1, 0, 2).reshape(m, hb * n)
        elif bias is not None and bias.dim() == 1:
            out = torch.addmm(bias, a, w)              # one launch (a @ w, then + bias, were two)
        else:
            out = a @ w
            if bias is not None:
                out = out + bias
        if init is not None:
            out = out + (init.repeat(1, hb) if batched and init.shape == (m, n) and hb > 1 else init)
        res = _apply_act(out, act, slope)
        if out_arg is not None:
            out_arg.copy_(res)
            return out_arg
        return res
    out = out_arg if out_arg is not None else torch.empty((m, hb * n), dtype=torch.float32, device=a.device)
    init_bs = n if batched else 0
    if init is not None and batched and init.shape == (m, n) and init.stride(-1) == 1:
        init_bs = 0                                  # one [M,N] init shared by all heads: batch stride 0
    elif init is not None and (init.stride(-1) != 1 or init.shape != out.shape):
        init = init.expand(m, hb * n).contiguous()
    if bias is not None:
        bias = bias.contiguous()
    if mode() == "f16x3":
        planes, b_scale = w_split if w_split is not None else split_weight_f16(w)
        if a_amax is None:
            a_amax = amax(a)
        # bench.py's PROFILE hook: the second field carries flops (2*M*N*K per head) for the "gemm_*" labels
        ops._launch("disgat_gemm_f16x3", "gemm_f16x3", 2.0 * m * n * k * hb,
                  a.data_ptr(), a.stride(-2), a.stride(0) if batched else 0, planes.data_ptr(),
                  a_amax.data_ptr(), b_scale.data_ptr(), ops._ptr(bias), ops._ptr(init),
                  0 if init is None else init.stride(0), init_bs, out.data_ptr(), out.stride(0), n if batched else 0,
                  m, n, k, hb, act, float(slope), ops._stream())
        return out
    planes = split_weight(w)
    _lib.call("disgat_gemm_split", a.data_ptr(), a.stride(-2), a.stride(0) if batched else 0, planes.data_ptr(),
              ops._ptr(bias), ops._ptr(init), 0 if init is None else init.stride(0), init_bs,
              out.data_ptr(), out.stride(0), n if batched else 0, m, n, k, hb, act, float(slope),
              3 if mode() == "split3" else 6, ops._stream())
    return out


def _act_backward(g, out, act, slope):
    """(g * act'(.), max |result| or None) from the saved output in one pass (csrc/gemm_split.hip: act_bwd_kernel)."""
    g = g.contiguous()
    if not (g.is_cuda and out.is_contiguous() and g.numel() % 4 == 0 and g.data_ptr() % 16 == 0):
        neg = out + 1.0 if act == ACT_ELU else torch.full_like(out, slope)
        return g * torch.where(out > 0, torch.ones_like(out), neg), None
    gin = torch.empty_like(g)
    am = torch.empty(1, dtype=torch.float32, device=g.device)
    _lib.call("disgat_act_bwd", g.data_ptr(), out.data_ptr(), gin.data_ptr(), g.numel(), act, float(slope), am.data_ptr(),
              ops._stream())
    return gin, am


def _tn_ok(a, g, k, n):
    return (mode() == "f16x3" and a.is_cuda and a.dtype == torch.float32 and g.dtype == torch.float32 and k % 128 == 0
            and n % 128 == 0 and a.stride(-1) == 1 and g.stride(-1) == 1 and a.stride(-2) % 4 == 0 and g.stride(-2) % 4 == 0
            and a.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0
            and (a.dim() == 2 or (a.stride(0) % 4 == 0 and g.stride(0) % 4 == 0)) and a.shape[-2] >= 4096)


def _weight_grad(a, g, a_amax, g_amax):
    """a^T @ g reducing over the rows (a [M,K] / [H,M,K], g [M,N] / [H,M,N] view) on the split-K f16x3 kernel."""
    batched = a.dim() == 3
    hb = a.shape[0] if batched else 1
    m, k = a.shape[-2:]
    n = g.shape[-1]
    tiles = (k // 128) * (n // 128) * hb
    splits = max(1, min(64, -(-1024 // tiles), m // 2048))
    if splits >= 8:
        splits -= splits % 8        # whole splits per XCD (the kernel's block map keeps a split's tiles on one L2)
    part = torch.empty((hb, splits, k, n), dtype=torch.float32, device=a.device)
    _lib.call("disgat_gemm_f16x3_tn", a.data_ptr(), a.stride(-2), a.stride(0) if batched else 0, g.data_ptr(), g.stride(-2),
              g.stride(0) if batched else 0, a_amax.data_ptr(), g_amax.data_ptr(), part.data_ptr(), m, k, n, hb, splits,
              ops._stream())
    out = part.sum(1) if splits > 1 else part[:, 0]
    return out if batched else out[0]


def _wgrad_small_ok(a, g, m, k, n, hb):
    """Shapes disgat_wgrad_small is for: a result of at most 256 K floats (a few 64 x 64 tiles: the nhid = 64 layers) reduced
    over at least 512 rows; fp32, unit inner strides, everything 16-byte aligned.  DISGAT_WGRAD_SMALL=0 switches it off."""
    if os.environ.get("DISGAT_WGRAD_SMALL", "1") == "0" or mode() == "blas":
        return False
    ok = (a.is_cuda and a.dtype == torch.float32 and g.dtype == torch.float32 and m >= 512 and hb * k * n <= 262144
          and k % 4 == 0 and n % 4 == 0 and a.stride(-1) == 1 and g.stride(-1) == 1 and a.stride(-2) % 4 == 0 and g.stride(-2) % 4 == 0
          and a.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0)
    return ok and (a.dim() == 2 or (a.stride(0) % 4 == 0 and g.stride(0) % 4 == 0))


def _wgrad_small(a, g, m, k, n, hb):
    batched = a.dim() == 3
    tiles = hb * -(-k // 64) * -(-n // 64)
    splits = max(1, min(1024, 512 // tiles, m // 32))
    out = torch.empty((hb, k, n), dtype=torch.float32, device=a.device)
    part = torch.empty((hb, splits, k, n), dtype=torch.float32, device=a.device) if splits > 1 else None
    _lib.call("disgat_wgrad_small", a.data_ptr(), a.stride(-2), a.stride(0) if batched else 0, g.data_ptr(), g.stride(-2),
              g.stride(0) if batched else 0, m, k, n, hb, splits, ops._ptr(part), out.data_ptr(), ops._stream())
    return out if batched else out[0]


def _wgrad_blas(a, g):
    """a^T @ g (2-D, or head-batched [H, M, K] / [H, M, N] views) for the shapes outside the split-K kernel's tiling - the
    nhid = 64 layers of the bundled graphs.  The reduction runs over the M node rows while the result is a few 16 x 16 tiles:
    left as ONE GEMM, hipBLASLt gives it a handful of workgroups (77-430 us per call at M = 19 793, a quarter of cora_full's
    captured epoch).  Cut into row ranges it is a batched GEMM with enough tiles to fill the chip plus a fixed-order sum."""
    m, k, n = a.shape[-2], a.shape[-1], g.shape[-1]
    batched = a.dim() == 3
    hb = a.shape[0] if batched else 1
    if _wgrad_small_ok(a, g, m, k, n, hb):
        return _wgrad_small(a, g, m, k, n, hb)
    s = min(64, m // 256, max(1, 512 // max(1, hb * -(-k // 64) * -(-n // 64))))
    if m < 8192:
        s = 1         # Cora / chameleon-sized: the one GEMM takes ~10 us, the cut form four launches of a launch-bound step
    if s < 2 or not a.is_cuda:
        return torch.bmm(a.transpose(1, 2), g) if batched else a.t() @ g
    rows = m // s
    head = rows * s
    if batched:       # [H, M, K] views of [M, H, K] buffers: rows of a range are strided, the batch is (head, range)
        part = torch.matmul(a[:, :head].reshape(hb, s, rows, k).transpose(2, 3), g[:, :head].reshape(hb, s, rows, n)).sum(1)
        if head < m:
            part = part + torch.bmm(a[:, head:].transpose(1, 2), g[:, head:])
        return part
    part = torch.bmm(a[:head].reshape(s, rows, k).transpose(1, 2), g[:head].reshape(s, rows, n)).sum(0)
    if head < m:
        part = part + a[head:].t() @ g[head:]
    return part


def linear_backward(a, w, g, a_amax=None, need_a=True, need_w=True, g_amax=None, ga_init=None):
    """(grad a, grad w) of a @ w (2-D, no bias / activation) on the f16x3 kernels where their tiling allows.
    g_amax: max |g| (or an upper bound) as a device scalar when the producer of g measured it already;
    ga_init: a tensor of grad a's shape added in the GEMM epilogue (another contribution to the same gradient)."""
    g_am = None
    if mode() == "f16x3" and g.is_cuda and g.dim() == 2 and g.is_contiguous() and g.shape[1] % 4 == 0 and g.data_ptr() % 16 == 0:
        g_am = g_amax if g_amax is not None else amax(g)
    ga = gw = None
    if need_a:
        ga = _forward(g, w.t(), None, ga_init, ACT_NONE, 0.0, g_am)
    if need_w:
        if g_am is not None and _tn_ok(a, g, a.shape[1], g.shape[1]):
            gw = _weight_grad(a, g, a_amax if a_amax is not None else amax(a), g_am)
        else:
            gw = _wgrad_blas(a, g)
    return ga, gw


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, w, bias, init, act, slope, a_amax=None, w_split=None):
        out = _forward(a, w, bias, init, act, slope, a_amax, w_split)
        ctx.save_for_backward(a, w, out if act != ACT_NONE else None)
        ctx.a_amax = a_amax
        ctx.meta = (act, slope, bias is not None, init is not None)
        ctx.init_shared = init is not None and a.dim() == 3 and init.shape[1] != out.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        a, w, out = ctx.saved_tensors
        act, slope, has_bias, has_init = ctx.meta
        g_am = None
        if act != ACT_NONE:
            g, g_am = _act_backward(g, out, act, slope)
        ga = gw = gb = gi = None
        # grad of the data operand: the same split-bf16 GEMM with the transposed weight (falls back to hipBLASLt
        # through _forward's own shape check); the weight gradient reduces over the million-row dimension and
        # stays on hipBLASLt.
        # one max |g| pass serves both GEMMs below
        if g_am is None and (mode() == "f16x3" and g.is_cuda and g.dim() == 2 and g.is_contiguous() and g.shape[1] % 4 == 0
                             and g.data_ptr() % 16 == 0):
            g_am = amax(g)
        if a.dim() == 3:
            hb, m, k = a.shape
            g3 = g.view(m, hb, -1).permute(1, 0, 2)
            if ctx.needs_input_grad[0]:
                ga = _forward(g3, w.transpose(1, 2), None, None, ACT_NONE, 0.0, g_am).view(m, hb, k).permute(1, 0, 2)
            if ctx.needs_input_grad[1]:
                if g_am is not None and _tn_ok(a, g3, k, g3.shape[2]):
                    gw = _weight_grad(a, g3, ctx.a_amax if ctx.a_amax is not None else amax(a), g_am)
                else:
                    gw = _wgrad_blas(a, g3)
        else:
            if ctx.needs_input_grad[0]:
                ga = _forward(g, w.t(), None, None, ACT_NONE, 0.0, g_am)
            if ctx.needs_input_grad[1]:
                if g_am is not None and _tn_ok(a, g, a.shape[1], g.shape[1]):
                    gw = _weight_grad(a, g, ctx.a_amax if ctx.a_amax is not None else amax(a), g_am)
                else:
                    gw = _wgrad_blas(a, g)
        if has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        if has_init and ctx.needs_input_grad[3]:
            gi = g.view(g.shape[0], a.shape[0], -1).sum(1) if ctx.init_shared else g
        return ga, gw, gb, gi, None, None, None, None


def linear(a, w, bias=None, init=None, act=ACT_NONE, slope=0.01, a_amax=None, w_split=None):
    """act(a @ w + bias + init); see module docstring for the batched form (bias then is [H*N]).
    a_amax: optional precomputed amax(a) (or an upper bound) when the same operand feeds several GEMMs;
    w_split: optional presplit(w) to reuse."""
    return _Linear.apply(a, w, bias, init, act, slope, a_amax, w_split)


class _SkinnyLinear(torch.autograd.Function):
    """x @ w.T + b for a handful of output columns (disgat_linear_skinny); backward = the plain matmuls."""

    @staticmethod
    def forward(ctx, x, w, b):
        y = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
        _lib.call("disgat_linear_skinny", x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], w.data_ptr(), w.stride(0),
                  0 if b is None else b.data_ptr(), w.shape[0], y.data_ptr(), y.stride(0), ops._stream())
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            g = g.contiguous()
            n_waves = int(min(4096, max(4, (x.shape[0] + 63) // 64 // 4 * 4)))
            part = torch.empty((n_waves, w.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
            _lib.call("disgat_linear_skinny_wgrad", x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], g.data_ptr(), g.stride(0),
                      w.shape[0], part.data_ptr(), n_waves, ops._stream())
            gw = part.sum(0)
        gb = g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


def skinny_ok(x, lin):
    """A torch.nn.Linear the skinny kernel takes: fp32 device rows of 256 / 512 columns, at most 16 outputs."""
    w = lin.weight
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and w.dtype == torch.float32 and w.is_cuda
            and x.shape[1] in (256, 512) and 0 < w.shape[0] <= 16 and x.stride(1) == 1 and x.stride(0) % 4 == 0
            and w.stride(1) == 1 and w.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and (lin.bias is None or lin.bias.is_contiguous()))


def skinny_linear(x, lin):
    """lin(x) for a skinny output layer on disgat_linear_skinny (HBM-bound: one pass over x)."""
    return _SkinnyLinear.apply(x, lin.weight, lin.bias)
