"""disgat_seg_tables (csrc/seg_tables.hip: the fixed-capacity work-item tables of the backward's segment passes, built on the
device for steps captured in a HIP graph) against graph.build_items - the host-side builder of the eager path, which reads
sizes back: same slices, same slot numbering, same split-key tables; plus the padding conventions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["rows", "cols", "no_slices", "hub", "empty_keys"])
def test_tables_equal_the_host_builder(case):
    from edgedisentangle_ssl_amd import ops_bwd
    from edgedisentangle_ssl_amd.graph import build_items
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    n_keys, chunk = 300, 16
    if case == "hub":
        keys = torch.cat([torch.randint(0, n_keys, (3000,), generator=g), torch.full((1500,), 17), torch.full((40,), n_keys - 1)])
    elif case == "empty_keys":
        keys = torch.randint(100, 140, (900,), generator=g)
    elif case == "no_slices":
        keys, chunk = torch.randint(0, n_keys, (200,), generator=g), 512
    else:
        keys = torch.randint(0, n_keys, (5000,), generator=g)
    if case != "cols":
        keys = torch.sort(keys).values
    keys = keys.to(dev)
    wi, perm, perm32 = ops_bwd._segments_static(keys, n_keys, case != "cols", chunk)
    c_len = keys.numel()
    if case == "cols":
        want = torch.sort(keys.to(torch.int32), stable=True).indices
        assert torch.equal(perm, want) and torch.equal(perm32.long(), want)
        keys = keys[perm]
    else:
        assert perm is None and perm32 is None
    ptr = torch.searchsorted(keys, torch.arange(n_keys + 1, device=dev))
    ref = build_items(ptr, chunk if case != "no_slices" else 10 ** 9)
    items = wi.items.cpu().numpy()
    tot = wi.totals.cpu().numpy()
    assert tot[0] == ref.n_items and wi.n_items == items.shape[0] >= ref.n_items
    assert np.all(items[tot[0]:] == np.array([-1, 0, 0, -1]))                      # padding items
    mine = items[: tot[0]]
    assert np.all(np.diff(mine[:, 0]) >= 0)                                         # key order
    theirs = ref.items.cpu().numpy()
    assert sorted(map(tuple, mine)) == sorted(map(tuple, theirs))
    if case == "no_slices":
        assert wi.n_split == 0 and wi.n_slots == 0 and np.all(mine[:, 3] == -1)
        return
    assert tot[1] == ref.n_split and tot[2] == ref.n_slots <= wi.n_slots and ref.n_split <= wi.n_split
    rows, sp = wi.split_rows.cpu().numpy(), wi.split_ptr.cpu().numpy()
    assert np.array_equal(rows[: ref.n_split], ref.split_rows.cpu().numpy()[: ref.n_split]) and np.all(rows[ref.n_split:] == -1)
    assert np.array_equal(sp[: ref.n_split + 1], ref.split_ptr.cpu().numpy()) and np.all(sp[ref.n_split:] == ref.n_slots)
    assert wi.n_items >= n_keys + c_len // chunk - 1
