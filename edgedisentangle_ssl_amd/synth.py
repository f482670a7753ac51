"""Synthetic power-law inputs of SURVEY 8(d) (BASELINE.json configs[2..4]).

  rng = numpy Generator(PCG64(1234)); m = (E-N)//2; w_i ~ (i+1)^-0.8; perm = rng.permutation(N)
  r = perm[rng.choice(N, m, p=w)];  c = rng.integers(0, N, m);  A = ((A0 + A0^T + I) > 0)
  features: torch.manual_seed(0); randn(N,F).   aux pairs: PCG64(99) uniform (2,M), row-major sorted,
  M_sup = floor(10*nnz/3); DisEdge lists split M_sup 1:3; labels = membership in A (and, for the
  homo / hetero lists, same / different class under a seeded 8-class node labelling).
`shard=(rank, world)` generates a statistically identical graph of world*N nodes whose rows are
drawn per rank (weak scaling: every rank owns N rows and ~E incident entries).
"""
import numpy as np
import torch

from . import sampling
from .graph import CSRGraph


def powerlaw_edges(n, e_target, seed=1234):
    rng = np.random.Generator(np.random.PCG64(seed))
    m = (e_target - n) // 2
    w = (np.arange(n, dtype=np.float64) + 1.0) ** -0.8
    w /= w.sum()
    perm = rng.permutation(n)
    r = perm[rng.choice(n, m, p=w)]
    c = rng.integers(0, n, m)
    return r.astype(np.int64), c.astype(np.int64)


def powerlaw_graph(n, e_target, device, seed=1234):
    r, c = powerlaw_edges(n, e_target, seed)
    r = torch.from_numpy(r).to(device)
    c = torch.from_numpy(c).to(device)
    loop = torch.arange(n, device=device)
    idx = torch.stack([torch.cat([r, c, loop]), torch.cat([c, r, loop])])
    return CSRGraph.from_index(idx, n)


def features(n, f, device, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(n, f, generator=g).to(device)


def node_labels(n, device, classes=8, seed=7):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.integers(0, classes, n)).to(device)


def uniform_pairs(n, m, device, seed=99):
    """Row-major sorted uniform pairs (duplicates kept, like independent draws)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    flat = torch.from_numpy(rng.integers(0, n * n, m)).to(device)
    flat = torch.sort(flat).values
    rows = torch.div(flat, n, rounding_mode="floor")
    return torch.stack([rows, flat - rows * n]), flat


def ssl_lists(graph, labels, seed=99):
    """(sup_idx, sup_lab), (homo_idx, homo_lab), (het_idx, het_lab) for the three SSL losses."""
    n = graph.n
    m_sup = (10 * graph.nnz) // 3
    pos = sampling.flat_edges(graph)
    same = labels[graph.row] == labels[graph.col.long()]
    sup_idx, sup_flat = uniform_pairs(n, m_sup, graph.device, seed)
    ho_idx, ho_flat = uniform_pairs(n, m_sup // 4, graph.device, seed + 1)
    he_idx, he_flat = uniform_pairs(n, m_sup - m_sup // 4, graph.device, seed + 2)
    return ((sup_idx, sampling.membership(sup_flat, pos)),
            (ho_idx, sampling.membership(ho_flat, pos[same])),
            (he_idx, sampling.membership(he_flat, pos[~same])))
