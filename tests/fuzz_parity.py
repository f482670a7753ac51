#!/usr/bin/env python3
"""Randomised parity sweep (GPU): random graphs / head counts / widths / attention + aggregation
types / aux lists / head ranges / chunk sizes / dropout-free forward + backward against the
float64 CPU oracle.  Prints every failing configuration; exit code 1 if any."""
import argparse
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs_common as ic  # noqa: E402
import kink  # noqa: E402
import edgedisentangle_ssl_amd as pkg  # noqa: E402
from edgedisentangle_ssl_amd import ops  # noqa: E402
from oracle import disgat_oracle as orc  # noqa: E402


def close(a, b, tol, what):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if b.numel() == 0:
        return
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert torch.isfinite(a).all(), f"{what}: non-finite"
    assert err <= tol * scale, f"{what}: {err:.3e} > {tol:.0e}*{scale:.3e}"


def one(case, rng):
    dev = torch.device("cuda")
    n = int(rng.integers(3, 400))
    H = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 20]))
    f_in = int(rng.choice([1, 3, 4, 7, 16, 33, 64, 100, 255, 256, 257, 300, 520, 700, 1433]))
    f_out = int(rng.choice([1, 2, 5, 8, 16, 31, 32, 64, 96, 130, 256]))
    att = int(rng.choice([1, 2, 3]))
    gnn = str(rng.choice(["AT", "SAGE", "GCN"]))
    chunk = int(rng.choice([2, 5, 16, 128, 1 << 20]))
    ne = int(rng.integers(0, n * 8))
    idx = torch.from_numpy(np.stack([rng.integers(0, n, ne), rng.integers(0, n, ne)]).astype(np.int64))
    if rng.random() < 0.5 and n > 4:                          # a hub row and a hub column
        hub = torch.from_numpy(np.stack([np.full(n, 1), np.arange(n)]).astype(np.int64))
        idx = torch.cat([idx, hub, hub.flip(0)], 1)
    ci = ic.coalesced_index_set(idx, n) if idx.shape[1] else idx
    if ci.shape[1] == 0:
        return "skip (oracle undefined on empty edge list)"
    naux = int(rng.integers(0, 3))
    aux = [torch.from_numpy(np.stack([rng.integers(0, n, m), rng.integers(0, n, m)]).astype(np.int64))
           for m in (int(rng.integers(0, 300)) for _ in range(naux))]
    if aux and rng.random() < 0.5:
        aux[0] = aux[0][:, torch.argsort(aux[0][0] * n + aux[0][1])]         # row-major like the reference sampler
    ranges = None
    if aux and rng.random() < 0.5:
        ranges = []
        for _ in aux:
            lo = int(rng.integers(0, H))
            ranges.append((lo, int(rng.integers(lo + 1, H + 1))))
    desc = f"case {case}: n={n} E={ci.shape[1]} H={H} F_in={f_in} F_out={f_out} att={att} gnn={gnn} chunk={chunk} aux={[a.shape[1] for a in aux]} ranges={ranges}"
    ops.CHUNK = {1: chunk, 2: chunk, 3: chunk, 4: chunk}
    layers = [ic.load_params(pkg.DisGALayer(f_in, f_out, dropout=0.0, alpha=0.1, att_type=att, gnn_type=gnn), 900 + 7 * case + h)
              .to(dev).eval() for h in range(H)]
    x = torch.from_numpy(rng.standard_normal((n, f_in)).astype(np.float32) * 0.5)
    adj = torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1]), (n, n)).to(dev)
    xg = x.to(dev).requires_grad_(True)
    recorded = []
    with kink.record_operands(recorded):
        heads, e_list, aux_out = pkg.disga_heads(layers, xg, adj, [a.to(dev) for a in aux] if aux else None, ranges)
    # att 3: the oracle's gradient follows the kernels' side wherever a leaky-ReLU argument is within 1e-5 of the kink
    # (tests/kink.py) - no case is redrawn or skipped for it
    pins = kink.Pins(recorded, H, f_out, [(ci[0], ci[1])] + [(a_[0], a_[1]) for a_ in aux]) if att == 3 else None
    wh = torch.from_numpy(rng.standard_normal((H, n, f_out)))
    we = torch.from_numpy(rng.standard_normal((H, ci.shape[1])) * 0.1)
    loss = sum((heads[h].double() * wh[h].to(dev)).sum() + (e_list[h][:, 0].double() * we[h].to(dev)).sum() for h in range(H))
    wa = [torch.from_numpy(rng.standard_normal((H, a.shape[1])) * 0.1) for a in aux]
    for h in range(H):
        for li in range(len(aux)):
            t = aux_out[h][li]
            lo, hi = (0, H) if ranges is None else ranges[li]
            assert (t is not None) == (lo <= h < hi), desc
            if t is not None and t.numel():
                loss = loss + (t[:, 0].double() * wa[li][h].to(dev)).sum()
    loss.backward()
    xc = x.double().requires_grad_(True)
    ref_loss = 0.0
    sds = []
    for h, lay in enumerate(layers):
        sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in lay.state_dict().items()}
        sds.append(sd)
        with kink.pinned_oracle(pins):
            ho, e, au = orc.disga_layer(xc, ci, sd, att, gnn, aux if aux else None)
        close(heads[h], ho, 1e-4, desc + f" head {h}")
        close(e_list[h][:, 0], e[:, 0], 1e-4, desc + f" edge_e {h}")
        ref_loss = ref_loss + (ho * wh[h]).sum() + (e[:, 0] * we[h]).sum()
        for li in range(len(aux)):
            lo, hi = (0, H) if ranges is None else ranges[li]
            if lo <= h < hi:
                close(aux_out[h][li][:, 0], au[li][:, 0], 1e-4, desc + f" aux{li} {h}")
                ref_loss = ref_loss + (au[li][:, 0] * wa[li][h]).sum()
    ref_loss.backward()
    if pins is not None:
        assert pins.disagree_far == 0, desc + f": {pins.disagree_far} sign disagreements away from the kink"
    gtol = 3e-4
    close(xg.grad, xc.grad, gtol, desc + " grad x")
    for h, lay in enumerate(layers):
        for k, prm in lay.named_parameters():
            want = sds[h][k].grad if sds[h][k].grad is not None else torch.zeros_like(sds[h][k])
            got = prm.grad if prm.grad is not None else torch.zeros_like(prm)
            close(got, want, gtol, desc + f" grad head{h}.{k}")
    return "ok"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    o = ap.parse_args()
    rng = np.random.Generator(np.random.PCG64(o.seed))
    fails = 0
    default_chunk = dict(ops.CHUNK)
    for c in range(o.cases):
        try:
            r = one(c, rng)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print("FAIL", str(e)[:400])
            if not isinstance(e, AssertionError):
                traceback.print_exc()
        if c % 25 == 24:
            print(f"[fuzz] {c + 1} cases, {fails} failures", flush=True)
    ops.CHUNK = default_chunk
    print(f"[fuzz] done: {o.cases} cases, {fails} failures")
    sys.exit(1 if fails else 0)
