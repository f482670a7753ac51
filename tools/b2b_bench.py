#!/usr/bin/env python3
"""Same-box timing of disgat_proj_fuse (csrc/gemm_b2b.hip) against the two-launch plane chain it replaces
(projection planes -> planes, fuser planes -> fp32), interleaved rounds, HIP events on the launch stream.

  python tools/b2b_bench.py [--m 1000000] [--heads 8] [--feat 256] [--rounds 5] [--gcn]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import ops_gemm as og  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1_000_000)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--feat", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--gcn", action="store_true")
    o = ap.parse_args()
    dev = torch.device("cuda")
    m, H, F = o.m, o.heads, o.feat
    g = torch.Generator(device="cuda").manual_seed(0)
    z = torch.randn(m, H, F, device=dev, generator=g)
    w1 = torch.randn(H, F, F, device=dev, generator=g) * (1.4 / F ** 0.5)
    w2 = torch.randn(H * F, F, device=dev, generator=g) * (1.0 / (H * F) ** 0.5)
    b1 = torch.randn(H * F, device=dev, generator=g) * 0.1 if o.gcn else None
    b2 = torch.randn(F, device=dev, generator=g) * 0.1
    zp = og.split_planes(z.permute(1, 0, 2))
    del z
    bound = torch.clamp(zp.bound * w1.abs().sum(1).max() * 1.001 + (b1.abs().max() if o.gcn else 0.0), min=1.0).reshape(1)
    w1r, w2r = og.presplit_rm(w1), og.presplit_rm(w2)
    wch = og.presplit_b2b(w1, w2)

    def b2b():
        return og.proj_fuse(zp, wch, b1, b2, bound, F, F, og.ACT_LEAKY, 0.01)

    def chain():
        _, hp = og.linear_planes(zp, w1r, F, b1, None, og.ACT_ELU, 0.0, False, bound)
        return og.linear_planes(hp, w2r, F, b2, None, og.ACT_LEAKY, 0.01)[0]

    def timed(fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(o.reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / o.reps

    a, c = b2b(), chain()
    torch.cuda.synchronize()
    sc = float(c.abs().max())
    print(f"max |b2b - chain| / max |chain| = {float((a - c).abs().max()) / sc:.3e}", flush=True)
    del a, c
    tb, tc = [], []
    for _ in range(o.rounds):
        tb.append(timed(b2b))
        tc.append(timed(chain))
    flops = 2.0 * m * H * F * (F + F) * 3
    print(f"M={m} H={H} F={F}: b2b {min(tb):.3f} ms (rounds {', '.join(f'{t:.3f}' for t in tb)}) = {flops / min(tb) / 1e9:.0f} TFLOP/s of fp16 MFMA; "
          f"chain {min(tc):.3f} ms ({', '.join(f'{t:.3f}' for t in tc)})", flush=True)


if __name__ == "__main__":
    main()
