"""Seeded surrogate node features for graphs whose feature matrix is not shipped (the reference's
cora / cora_full `feature_new.npy` are listed in .MISSING_LARGE_BLOBS): non-negative, sparse,
row-sum normalised like data_load.py:137-144.  Same generator as tests/inputs_common.features."""
import zlib

import numpy as np
import torch


def surrogate_features(n, f, seed=51):
    g = np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(b"features_cora_surrogate")]))
    x = (g.random((n, f)) < 0.15) * g.random((n, f))
    x[np.arange(n), g.integers(0, f, n)] += 0.5
    x = x / x.sum(1, keepdims=True)
    return torch.from_numpy(x.astype(np.float32))
