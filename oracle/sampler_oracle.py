"""CPU restatement (numpy) of the SSL pair sampler - TEST INFRASTRUCTURE, never imported by the product path.

What it restates: SupEdgeTrainer.sample_train / GeneratedEdgeTrainer.sample_train of the reference
(/root/reference/pretrainer.py:683-707, 524-576):

    mask = (rand(N, N) < 3 rho)  |  {the first third of the shuffled positives};   indices = mask.nonzero().T  (row-major)
    labels = (adj != 0) at those entries

in the O(output) formulation of edgedisentangle_ssl_amd/csrc/pair_sample.hip, draw for draw: the same counter-based
generator (Philox2x32-10 keyed by splitmix64(seed, step)), the same geometric skipping along column intervals with the
same explicit double-precision logarithm, the same Feistel permutation selecting exactly n_pos // 3 positives.  The
outputs are integers, so the GPU tests demand EQUALITY with this file (tests/test_gpu_sampler.py); the distribution itself
is pinned against the reference in tests/test_sampler_stats.py (closed-form moments of the reference's mask and the
reference's own recorded draw, tests/golden/tiny_ref_sampler.npz - the reference's RNG stream cannot be matched entry for
entry by any O(E) sampler).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math

import numpy as np

PCAP, RCAP, RMEAN = 256, 256, 96          # include/disgat_hip.h: DISGAT_SAMPLE_*
FEISTEL_ROUNDS = 8
_M64 = (1 << 64) - 1
_M32 = np.uint64(0xFFFFFFFF)


def splitmix64(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def make_keys(seed, step):
    m = splitmix64((seed + step * 0x9E3779B97F4A7C15) & _M64)
    fk = [splitmix64((m + r + 1) & _M64) & 0xFFFFFFFF for r in range(FEISTEL_ROUNDS)]
    return m & 0xFFFFFFFF, m >> 32, fk


def fmix32(h):
    """murmur3 finaliser on uint64 arrays holding 32-bit values."""
    h = h ^ (h >> np.uint64(16))
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h = h ^ (h >> np.uint64(13))
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    return h ^ (h >> np.uint64(16))


def philox_u64(c0, c1, key):
    """Philox2x32-10; counters are uint64 arrays holding 32-bit values.  Returns (word0 << 32) | word1."""
    c0, c1 = (np.asarray(c, dtype=np.uint64) for c in np.broadcast_arrays(c0, c1))
    key = int(key)
    for _ in range(10):
        p = np.uint64(0xD256D193) * c0
        c0 = (p >> np.uint64(32)) ^ np.uint64(key) ^ c1
        c1 = p & _M32
        key = (key + 0x9E3779B9) & 0xFFFFFFFF
    return (c0 << np.uint64(32)) | c1


def det_log(u):
    """ln(u), u in (0, 1], by the kernel's operation sequence (every numpy operation rounds once, as the kernel does with
    floating-point contraction off)."""
    m, e = np.frexp(u)
    low = m < 0.70710678118654752440
    m = np.where(low, m * 2.0, m)
    e = np.where(low, e - 1, e)
    s = (m - 1.0) / (m + 1.0)
    z = s * s
    p = np.full_like(z, 1.0 / 15.0)
    for k in (13, 11, 9, 7, 5, 3):
        p = p * z + 1.0 / k
    p = p * z + 1.0
    lm = (2.0 * s) * p
    return e.astype(np.float64) * 0.69314718055994530942 + lm


def feistel_domain(n_pos):
    """(ka, b): the permutation lives on Z_(2^ka) x Z_b, 2^ka ~ sqrt(n_pos), b = ceil(n_pos / 2^ka)."""
    bits = 0
    while (1 << bits) < n_pos:
        bits += 1
    ka = bits // 2
    return ka, (n_pos + (1 << ka) - 1) >> ka if n_pos > 0 else 1


def selected(j, fk, n_pos, n_sel):
    """j: uint64 array of positive ids < n_pos -> bool: pi(j) < n_sel."""
    ka, b = feistel_domain(n_pos)
    amask, ub = np.uint64((1 << ka) - 1), np.uint64(b)
    v = np.asarray(j, dtype=np.uint64).copy()
    todo = np.ones(v.shape, dtype=bool)
    while todo.any():
        w = v[todo]
        left, right = w & amask, w >> np.uint64(ka)
        for r in range(0, FEISTEL_ROUNDS, 2):
            left = left ^ (fmix32((right + np.uint64(fk[r])) & _M32) & amask)
            right = right + ((fmix32((left + np.uint64(fk[r + 1])) & _M32) * ub) >> np.uint64(32))
            right = np.where(right >= ub, right - ub, right)
        w = (right << np.uint64(ka)) | left
        v[todo] = w
        todo[todo] = w >= n_pos
    return v < n_sel


def build_items(rowptr, col, n_rows, n_cols, p):
    """int32 [n_items, 8] = {row, col_lo, col_hi, pos_lo, pos_hi, 0, 0, 0}: every row's column range cut at the multiples of
    lblk = floor(RMEAN / p) (expected random entries per interval <= RMEAN) and at every PCAP-th positive of the row."""
    lblk = n_cols if p <= 0 else min(n_cols, max(1, int(RMEAN / p)))
    items = []
    for r in range(n_rows):
        lo, hi = int(rowptr[r]), int(rowptr[r + 1])
        cuts = set(range(0, n_cols, lblk))
        cuts.update(int(col[j]) for j in range(lo + PCAP, hi, PCAP))
        cuts = sorted(cuts) + [n_cols]
        cols = np.asarray(col[lo:hi], dtype=np.int64)
        for a, b in zip(cuts[:-1], cuts[1:]):
            items.append((r, a, b, lo + int(np.searchsorted(cols, a)), lo + int(np.searchsorted(cols, b)), 0, 0, 0))
    return np.asarray(items, dtype=np.int32).reshape(-1, 8)


def sample(items, pos_col, n_sel, p, seed, step, n_cols):
    """One list.  Returns (rows int64 [M], cols int64 [M], labels float32 [M], n_over) - row-major sorted."""
    items = np.asarray(items, dtype=np.int64)
    pos_col = np.asarray(pos_col, dtype=np.int64)
    n_pos = int(pos_col.shape[0])
    k0, k1, fk = make_keys(int(seed), int(step))
    n_items = items.shape[0]
    row, clo, chi, plo, phi = (items[:, i] for i in range(5))
    n_over = 0
    # ---- selected positives
    pos_row = np.repeat(row, phi - plo)
    assert pos_row.shape[0] == n_pos, "the items must cover every positive exactly once, in order"
    sel = selected(np.arange(n_pos, dtype=np.uint64), fk, n_pos, n_sel) if n_sel > 0 else np.zeros(n_pos, dtype=bool)
    # ---- Bernoulli(p) process per item, 64 draws per round
    r_item, r_col = [], []
    if p > 0:
        inv = 1.0 / math.log1p(-p) if p < 1 else -0.0
        active = np.nonzero(chi > clo)[0]
        first = clo[active].copy()
        cap = (chi - clo + 1)[active]
        n_r = np.zeros(active.shape[0], dtype=np.int64)
        lane = np.arange(64, dtype=np.uint64)
        rnd = 0
        while active.size:
            bits = philox_u64(active.astype(np.uint64)[:, None] ^ np.uint64(k1), np.uint64(rnd * 64) + lane[None, :], k0)
            u = ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
            with np.errstate(invalid="ignore"):
                g = np.floor(det_log(u) * inv)
            capf = (cap - 1).astype(np.float64)[:, None]
            inc = np.where(g >= capf, cap[:, None], g.astype(np.int64) + 1)
            incl = np.minimum(np.cumsum(inc, axis=1), cap[:, None])
            valid = incl <= (chi[active] - first)[:, None]
            x = first[:, None] - 1 + incl
            nv = valid.sum(1)
            room = RCAP - n_r
            over = nv > room
            n_over += int(over.sum())
            keep = valid & (np.arange(64)[None, :] < room[:, None])
            ii, ll = np.nonzero(keep)
            r_item.append(active[ii])
            r_col.append(x[ii, ll])
            n_r = n_r + np.minimum(nv, room)
            cont = (nv == 64) & ~over & (x[:, 63] + 1 < chi[active])
            first = (x[:, 63] + 1)[cont]
            active, cap, n_r = active[cont], cap[cont], n_r[cont]
            rnd += 1
    r_item = np.concatenate(r_item) if r_item else np.zeros(0, dtype=np.int64)
    r_col = np.concatenate(r_col) if r_col else np.zeros(0, dtype=np.int64)
    flat_r = row[r_item] * n_cols + r_col
    flat_pos = pos_row * n_cols + pos_col
    flat = np.union1d(flat_r, flat_pos[sel])
    labels = np.isin(flat, flat_pos).astype(np.float32)
    return flat // n_cols, flat % n_cols, labels, n_over


def padded(rows, cols, labels, capacity, n_rows, n_cols):
    """The fixed-capacity form: (idx int64 [2, capacity], labels [capacity]) with padding pair (n_rows-1, n_cols-1), label -1."""
    m = min(int(rows.shape[0]), capacity)
    idx = np.empty((2, capacity), dtype=np.int64)
    lab = np.full(capacity, -1.0, dtype=np.float32)
    idx[0, :m], idx[1, :m], lab[:m] = rows[:m], cols[:m], labels[:m]
    idx[0, m:], idx[1, m:] = n_rows - 1, n_cols - 1
    return idx, lab
