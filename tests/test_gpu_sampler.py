"""The pair sampler kernels (csrc/pair_sample.hip through sampling.PairSampler and the C ABI) against the numpy restatement
oracle/sampler_oracle.py: the outputs are integers (and 0/1 labels), so the bar is EQUALITY - every index, every label,
the list length, over several generator steps, in the exact-length and the fixed-capacity (padded) form.  Cases: the tiny
graph, the bundled real graphs (SupEdge's positives = all entries, DisEdge's = the same-label / different-label subsets),
a hub row of 1 000 positives (items cut inside a row), a dense mask (several column blocks per row; random columns that
coincide with positives), a row shard (more columns than rows, global density), no positives, p = 0 and p = 1.
Then properties at BASELINE's full size (1M nodes / 20M entries, configs[3]) and inside a captured HIP graph."""
import os

import numpy as np
import pytest
import torch

import inputs_common as ic

pytestmark = pytest.mark.gpu


def _oracle_lists(smp, seed, steps):
    from oracle import sampler_oracle as so
    items = smp.items.cpu().numpy()
    col = smp.pos_col.cpu().numpy()
    return [so.sample(items, col, smp.n_sel, smp.p, seed, s, smp.n_cols) for s in range(steps)]


def _check_equal(n_rows, pos, n_cols=None, n_pos_global=None, steps=3, seed=2024, static=True):
    from edgedisentangle_ssl_amd import sampling
    from oracle import sampler_oracle as so
    dev = torch.device("cuda:0")
    pos_t = torch.from_numpy(np.asarray(pos, dtype=np.int64)).to(dev)
    smp = sampling.PairSampler(n_rows, pos_t, n_cols=n_cols, n_pos_global=n_pos_global, seed=seed)
    rowptr = np.searchsorted(pos, np.arange(n_rows + 1, dtype=np.int64) * smp.n_cols)
    assert np.array_equal(smp.items.cpu().numpy(), so.build_items(rowptr, np.asarray(pos) % smp.n_cols, n_rows, smp.n_cols, smp.p))
    want = _oracle_lists(smp, seed, steps + (steps if static else 0))
    for s in range(steps):
        idx, lab = smp.sample()
        r, c, l, over = want[s]
        assert over == 0
        assert idx.shape == (2, r.shape[0]), (s, idx.shape, r.shape)
        assert np.array_equal(idx[0].cpu().numpy(), r) and np.array_equal(idx[1].cpu().numpy(), c), s
        assert np.array_equal(lab.cpu().numpy(), l), s
        assert idx._disgat_checked[:2] == (n_rows, smp.n_cols)
    if static:
        for s in range(steps, 2 * steps):
            idx, lab = smp.sample_static()
            r, c, l, _over = want[s]
            pi, pl = so.padded(r, c, l, smp.capacity, n_rows, smp.n_cols)
            assert float(lab._disgat_count) == min(r.shape[0], smp.capacity)
            assert np.array_equal(idx.cpu().numpy(), pi) and np.array_equal(lab.cpu().numpy(), pl), s
    assert smp.events() == (0, 0)
    assert int(smp.meta[1]) == (2 * steps if static else steps)
    return smp


def test_tiny_graph_equals_the_oracle():
    idx, _v, n = ic.tiny_graph()
    _check_equal(n, np.unique((idx[0] * n + idx[1]).numpy()), steps=8)


@pytest.mark.parametrize("name", ["chameleon", "cora", "cora_full"])
def test_real_graphs_equal_the_oracle(name, golden_dir):
    """SupEdge's positive set (every entry of the processed adjacency) and DisEdge's two (pretrainer.py:448-456)."""
    from edgedisentangle_ssl_amd import data_load
    from edgedisentangle_ssl_amd.graph import CSRGraph
    adj, _feat, labels = data_load.load_fixture(os.path.join(golden_dir, f"data_{name}.npz"))
    g = CSRGraph.from_adj(adj)
    n = g.n
    flat = (g.row * n + g.col.long()).numpy()
    lab = labels.numpy()
    same = lab[g.row.numpy()] == lab[g.col.numpy()]
    steps = 1 if name == "cora_full" else 2
    _check_equal(n, flat, steps=steps)
    _check_equal(n, flat[same], steps=steps, static=False)
    _check_equal(n, flat[~same], steps=steps, static=False)


def test_hub_rows_dense_masks_shards_and_corners():
    rng = np.random.default_rng(9)
    hub = np.unique(np.concatenate([7 * 3000 + rng.choice(3000, 1000, replace=False), 11 * 3000 + np.arange(3000),
                                    rng.integers(0, 500 * 3000, 4000)]))
    _check_equal(500, hub, n_cols=3000)                                               # rows of 1 000 and 3 000 positives: items cut inside a row
    _check_equal(50, np.unique(rng.integers(0, 2500, 900)))                           # p ~ 0.8
    _check_equal(30, np.unique(rng.integers(0, 30 * 2000, 20000)), n_cols=2000, n_pos_global=400000)   # p = 0.3, 7 column blocks per row
    _check_equal(64, np.unique(rng.integers(0, 64 * 512, 300)), n_cols=512, n_pos_global=2400)         # a shard: rows local, columns global
    _check_equal(40, np.zeros(0, dtype=np.int64), n_pos_global=80)                    # no positives here, the Bernoulli part alone
    _check_equal(40, np.unique(rng.integers(0, 1600, 90)), n_pos_global=0)            # p = 0: exactly the selected third
    _check_equal(24, np.arange(24 * 24, dtype=np.int64))                              # p = 1: every entry
    _check_equal(1, np.array([0], dtype=np.int64), n_cols=5, n_pos_global=3)


def test_seed_comes_from_torchs_generator_and_streams_differ():
    from edgedisentangle_ssl_amd import sampling
    dev = torch.device("cuda:0")
    pos = torch.unique(torch.randint(0, 300 * 300, (2000,), device=dev))
    torch.manual_seed(77)
    a = sampling.PairSampler(300, pos).sample()
    torch.manual_seed(77)
    b = sampling.PairSampler(300, pos).sample()
    c = sampling.PairSampler(300, pos).sample()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert a[0].shape != c[0].shape or not torch.equal(a[0], c[0])
    smp = sampling.PairSampler(300, pos, seed=5)
    first, second = smp.sample(), smp.sample()
    assert first[0].shape != second[0].shape or not torch.equal(first[0], second[0])          # the step advances on the device
    smp.reseed(5)
    again = smp.sample()
    assert torch.equal(first[0], again[0]) and torch.equal(first[1], again[1])


def test_list_over_capacity_is_clamped_and_counted():
    """A fixed-capacity list that comes out longer than the capacity (8 sigma: forced here) keeps its first `capacity`
    entries and raises the device-side event counter main.run reads."""
    from edgedisentangle_ssl_amd import sampling
    from oracle import sampler_oracle as so
    dev = torch.device("cuda:0")
    pos = torch.unique(torch.randint(0, 200 * 200, (900,), device=dev))
    smp = sampling.PairSampler(200, pos, seed=3)
    smp.capacity = 1000
    idx, lab = smp.sample_static()
    r, c, l, _ = so.sample(smp.items.cpu().numpy(), smp.pos_col.cpu().numpy(), smp.n_sel, smp.p, 3, 0, 200)
    assert r.shape[0] > 1000 and float(lab._disgat_count) == 1000 and smp.events() == (0, 1)
    assert np.array_equal(idx[0].cpu().numpy(), r[:1000]) and np.array_equal(idx[1].cpu().numpy(), c[:1000])
    assert np.array_equal(lab.cpu().numpy(), l[:1000])


def test_replays_of_a_captured_sampler_draw_fresh_lists():
    """The generator's step lives on the device and the plan advances it: a HIP graph holding plan + emit yields the oracle's
    list for step 0, 1, 2, ... on successive replays, with no host input."""
    from edgedisentangle_ssl_amd import sampling
    from oracle import sampler_oracle as so
    dev = torch.device("cuda:0")
    idx0, _v, n = ic.tiny_graph()
    pos = torch.unique(idx0[0] * n + idx0[1]).to(dev)
    smp = sampling.PairSampler(n, pos, seed=99)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        smp.sample_static()
    torch.cuda.current_stream().wait_stream(side)
    smp.reseed(99)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        idx, lab = smp.sample_static()
    items, col = smp.items.cpu().numpy(), smp.pos_col.cpu().numpy()
    for step in range(4):
        graph.replay()
        r, c, l, _ = so.sample(items, col, smp.n_sel, smp.p, 99, step, n)
        pi, pl = so.padded(r, c, l, smp.capacity, n, n)
        assert np.array_equal(idx.cpu().numpy(), pi) and np.array_equal(lab.cpu().numpy(), pl), step
        assert float(lab._disgat_count) == r.shape[0]


def test_full_size_list_properties():
    """BASELINE configs[3]'s graph (1M nodes / 20M entries): strictly row-major sorted, labels == membership, the length inside
    6 sigma of the closed form, exactly a third of the positives forced on (checked through the positives' hit count: a
    positive is listed with probability 1/3 + 2/3 p), per-row counts of the random part Binomial(N, p)."""
    from edgedisentangle_ssl_amd import sampling, synth
    dev = torch.device("cuda:0")
    n, e = 1_000_000, 20_000_000
    g = synth.powerlaw_graph(n, e, dev)
    pos = sampling.flat_edges(g)
    npos = int(pos.numel())
    smp = sampling.PairSampler(n, pos, seed=1)
    assert int((smp.items[:, 4] - smp.items[:, 3]).max()) <= sampling.PCAP
    p = smp.p
    for rep in range(2):
        idx, lab = smp.sample()
        flat = idx[0] * n + idx[1]
        assert bool(torch.all(flat[1:] > flat[:-1]))
        assert int(idx.min()) >= 0 and int(idx.max()) < n
        assert torch.equal(lab, sampling.membership(flat, pos))
        third = npos // 3
        mean = float(n) * n * p + third * (1 - p)
        sd = np.sqrt((float(n) * n - third) * p * (1 - p))
        assert abs(flat.numel() - mean) < 6 * sd, (flat.numel(), mean, sd)
        n_hit = int(lab.sum())
        mean_hit = third + (npos - third) * p
        assert abs(n_hit - mean_hit) < 6 * np.sqrt((npos - third) * p * (1 - p)) + 1, (n_hit, mean_hit)
        neg_rows = idx[0][lab == 0]
        per_row = torch.bincount(neg_rows, minlength=n).double()
        deg = (g.rowptr[1:] - g.rowptr[:-1]).double()
        expect = (n - deg) * p                                     # negatives of a row: Binomial(N - deg, p)
        z = float(((per_row - expect) ** 2 / expect.clamp(min=1e-9)).mean())
        assert 0.97 < z < 1.03, z                                  # index of dispersion ~ 1 - p over a million rows
        assert abs(float(per_row.sum()) - float(expect.sum())) < 6 * np.sqrt(float(expect.sum()))
    assert smp.events() == (0, 0)


def test_rank_share_of_the_8M_node_workload():
    """BASELINE configs[4] as one rank sees it: 1M local rows, column ids up to 8M, the GLOBAL density (8 ranks' 160M entries
    over 8M^2): the list is a row shard's - rows local, columns global, strictly row-major sorted, labels == membership in the
    shard's own entries, ~60 random entries per row, a third of the shard's positives forced on, no capacity event."""
    from edgedisentangle_ssl_amd import sampling
    dev = torch.device("cuda:0")
    n_rows, n_cols, per_row = 1_000_000, 8_000_000, 20
    g = torch.Generator(device=dev).manual_seed(3)
    rows = torch.arange(n_rows, device=dev).repeat_interleave(per_row)
    cols = torch.randint(0, n_cols, (n_rows * per_row,), device=dev, generator=g)
    pos = torch.unique(rows * n_cols + cols)
    npos = int(pos.numel())
    n_glob = 8 * npos
    smp = sampling.PairSampler(n_rows, pos, n_cols=n_cols, n_pos_global=n_glob, seed=4)
    assert smp.n_items >= n_rows and abs(smp.p - 3.0 * n_glob / float(n_cols) ** 2) < 1e-18
    idx, lab = smp.sample()
    flat = idx[0] * n_cols + idx[1]
    assert bool(torch.all(flat[1:] > flat[:-1]))
    assert int(idx[0].max()) < n_rows and int(idx[1].max()) < n_cols and int(idx[1].max()) > n_cols - 1000
    assert torch.equal(lab, sampling.membership(flat, pos))
    p, third = smp.p, npos // 3
    mean = float(n_rows) * n_cols * p + third * (1 - p)
    sd = np.sqrt(float(n_rows) * n_cols * p * (1 - p))
    assert abs(flat.numel() - mean) < 6 * sd, (flat.numel(), mean, sd)
    assert abs(int(lab.sum()) - (third + (npos - third) * p)) < 6 * np.sqrt((npos - third) * p) + 2
    assert smp.events() == (0, 0)
