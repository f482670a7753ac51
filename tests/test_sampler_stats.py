"""Distribution of the O(E) pair sampler (sampling.sample_pairs) against the reference's dense-mask sampler
(pretrainer.py:683-707, 524-576): mask = Bernoulli(3*rho) over all N^2 entries united with a third of the
positives.  The reference's own draw on the tiny graph is recorded in tests/golden/tiny_ref_sampler.npz; one draw
cannot be matched entry for entry (different RNG streams), so the test compares distributions:
  * M = K + n_pos//3 - overlap with K ~ Binomial(N^2, 3 rho): mean and spread of our M over many draws against the
    closed form, and the reference's recorded M inside that spread;
  * every positive appears with probability 1/3 + 2/3 * 3 rho, every negative with probability 3 rho;
  * the random part is uniform over rows (per-row counts are Binomial(N, 3 rho));
  * structure: row-major sorted, unique, labels == membership.
Runs on the CPU here and on the GPU (the sampler is device code only through torch ops)."""
import os

import numpy as np
import pytest
import torch

import inputs_common as ic


def _stats(dev, golden_dir, draws=600, static=False):
    from edgedisentangle_ssl_amd import sampling
    from edgedisentangle_ssl_amd.graph import CSRGraph
    idx, _vals, n = ic.tiny_graph()
    g = CSRGraph.from_index(idx.to(dev), n)
    pos = sampling.flat_edges(g)
    npos = int(pos.numel())
    gen = torch.Generator(device=dev).manual_seed(11)
    hgen = torch.Generator().manual_seed(12)
    p3 = 3.0 * npos / (n * n)
    ms, hit = [], torch.zeros(n * n, device=dev)
    smp = sampling.StaticSampler(n, pos) if static else None
    for _ in range(draws):
        if static:      # the fixed-capacity form captured steps use: valid prefix + padding (label -1, pair (n-1, n-1))
            smp.draw_k(hgen)
            pidx, lab = smp.sample(gen)
            c = int(lab._disgat_count)
            assert pidx.shape[1] == smp.capacity == lab.shape[0] and getattr(pidx, "_disgat_static", False)
            assert torch.all(lab[c:] == -1) and torch.all(pidx[:, c:] == n - 1)
            pidx, lab = pidx[:, :c], lab[:c]
        else:
            pidx, lab = sampling.sample_pairs(n, pos, gen, host_generator=hgen)
        flat = pidx[0] * n + pidx[1]
        assert torch.all(flat[1:] > flat[:-1])                       # mask.nonzero() order, no duplicates
        assert torch.equal(lab, torch.isin(flat, pos).float())
        assert int(pidx.min()) >= 0 and int(pidx.max()) < n
        hit[flat] += 1
        ms.append(int(flat.numel()))
    ms = np.asarray(ms, dtype=np.float64)
    third = npos // 3
    # E[M] = N^2 p3 + third * (1 - p3);  Var[M] = (N^2 - third) p3 (1 - p3)   (entries outside the third are Bernoulli)
    mean = n * n * p3 + third * (1 - p3)
    var = (n * n - third) * p3 * (1 - p3)
    if static:
        assert int(smp.short) == 0 and smp.clamped == 0
    assert abs(ms.mean() - mean) < 4 * np.sqrt(var / draws), (ms.mean(), mean)
    assert 0.8 * var < ms.var() < 1.25 * var, (ms.var(), var)
    ref = np.load(os.path.join(golden_dir, "tiny_ref_sampler.npz"))
    m_ref = ref["sup_idx"].shape[1]
    assert abs(m_ref - mean) < 4 * np.sqrt(var), (m_ref, mean)       # the reference's own draw is a typical one
    assert abs(ref["sup_lab"].mean() - (third + (npos - third) * p3) / mean) < 0.05
    freq = (hit / draws).cpu().numpy()
    is_pos = np.zeros(n * n, dtype=bool)
    is_pos[pos.cpu().numpy()] = True
    p_pos = third / npos + (1 - third / npos) * p3
    assert abs(freq[is_pos].mean() - p_pos) < 4 * np.sqrt(p_pos * (1 - p_pos) / (draws * is_pos.sum()))
    assert abs(freq[~is_pos].mean() - p3) < 4 * np.sqrt(p3 * (1 - p3) / (draws * (~is_pos).sum()))
    per_row = freq.reshape(n, n)[:, :].copy()
    per_row[is_pos.reshape(n, n)] = np.nan                           # negatives only: uniform over rows and columns
    rows = np.nanmean(per_row, 1)
    assert np.all(np.abs(rows - p3) < 6 * np.sqrt(p3 * (1 - p3) / (draws * (n - 8)))), rows
    cols = np.nanmean(per_row, 0)
    assert np.all(np.abs(cols - p3) < 6 * np.sqrt(p3 * (1 - p3) / (draws * (n - 24)))), cols


def test_sampler_distribution_cpu(golden_dir):
    _stats(torch.device("cpu"), golden_dir)


def test_static_sampler_distribution_cpu(golden_dir):
    _stats(torch.device("cpu"), golden_dir, static=True)


@pytest.mark.gpu
def test_static_sampler_distribution_gpu(golden_dir):
    _stats(torch.device("cuda:0"), golden_dir, static=True)


@pytest.mark.gpu
def test_sampler_distribution_gpu(golden_dir):
    _stats(torch.device("cuda:0"), golden_dir)


def test_binomial_count_and_subset():
    from edgedisentangle_ssl_amd import sampling
    g = torch.Generator().manual_seed(0)
    ks = np.array([sampling.binomial_count(10 ** 12, 6e-5, g) for _ in range(200)], dtype=np.float64)
    assert abs(ks.mean() - 6e7) < 4 * np.sqrt(6e7 / 200) and 0.7 * 6e7 < ks.var() < 1.4 * 6e7
    assert sampling.binomial_count(0, 0.5) == 0 and sampling.binomial_count(10, 0.0) == 0
    s = sampling.uniform_subset(50, 45, torch.device("cpu"), g)       # dense case: many collisions, still 45 distinct
    assert s.numel() == 45 and torch.all(s[1:] > s[:-1]) and int(s.max()) < 50
    assert sampling.uniform_subset(10, 99, torch.device("cpu"), g).numel() == 10


def test_sampler_on_a_row_shard():
    """Rows local, columns global (flat ids must use n_cols, not n_local)."""
    from edgedisentangle_ssl_amd import parallel, sampling
    from edgedisentangle_ssl_amd.graph import CSRGraph
    idx, _vals, n = ic.tiny_graph()
    g = CSRGraph.from_index(idx, n)
    full = set((g.row * n + g.col.long()).tolist())
    gen = torch.Generator().manual_seed(5)
    seen_pos = 0
    for rank in range(3):
        dg = parallel.DistGraph.shard(g, rank, 3)
        pos = sampling.flat_edges(dg)
        assert torch.all(pos[1:] > pos[:-1])
        pidx, lab = sampling.sample_pairs(dg.n, pos, gen, n_cols=dg.n_cols, n_pos_global=g.nnz)
        assert int(pidx[0].max()) < dg.n and int(pidx[1].max()) < n and int(pidx[1].max()) >= dg.n   # columns span all nodes
        glob = ((pidx[0] + dg.row_start) * n + pidx[1]).tolist()
        assert [float(f in full) for f in glob] == lab.tolist()
        seen_pos += int(lab.sum())
    assert seen_pos >= g.nnz // 3 - 3


def test_ranks_draw_their_own_streams_and_pool_to_the_unsharded_sample():
    """main.run reseeds every rank after the replicated set-up (main.reseed_rank): two equal-size shards must not score
    the same (local row, column) negatives, a rank's stream must be repeatable, and the lists pooled over the ranks
    must have the unsharded sample's size distribution (K ~ Binomial(N^2, 3 rho) random entries + a third of the
    positives)."""
    from edgedisentangle_ssl_amd import main as drop_in
    from edgedisentangle_ssl_amd import sampling
    n, world = 64, 2
    rng = np.random.default_rng(3)
    flat_all = np.unique(rng.integers(0, n * n, 150))
    npos = flat_all.size
    p3 = 3.0 * npos / (n * n)
    half = n // world
    shards = []
    for r in range(world):
        own = flat_all[(flat_all // n >= r * half) & (flat_all // n < (r + 1) * half)]
        shards.append(torch.from_numpy(own - r * half * n))            # local rows, global columns
    totals, same = [], 0
    for rep in range(200):
        lists = []
        for r in range(world):
            drop_in.reseed_rank(1000 + rep, r)
            idx, lab = sampling.sample_pairs(half, shards[r], n_cols=n, n_pos_global=npos)
            assert torch.equal(lab, torch.isin(idx[0] * n + idx[1], shards[r]).float())
            lists.append(idx)
        drop_in.reseed_rank(1000 + rep, 1)
        again, _ = sampling.sample_pairs(half, shards[1], n_cols=n, n_pos_global=npos)
        assert torch.equal(again, lists[1])                             # a rank's stream repeats under its seed
        neg = [set((i[0] * n + i[1]).tolist()) - set(s.tolist()) for i, s in zip(lists, shards)]
        same += len(neg[0] & neg[1]) / max(1, min(len(neg[0]), len(neg[1])))
        totals.append(sum(i.shape[1] for i in lists))
    assert same / 200 < 0.3, same / 200             # identical streams share ~all negatives, independent ones ~3 rho = 0.11
    third = sum(int(s.numel()) // 3 for s in shards)
    mean = n * n * p3 + third * (1 - p3)
    var = (n * n - third) * p3 * (1 - p3)
    t = np.asarray(totals, dtype=np.float64)
    assert abs(t.mean() - mean) < 4 * np.sqrt(var / t.size), (t.mean(), mean)
    assert 0.7 * var < t.var() < 1.4 * var, (t.var(), var)
