#!/usr/bin/env python3
"""disgat_gemm_planes on the DISGAT shapes: error vs float64 at small M, time at M = 1e6 beside the fp32-input kernels.
tools/gemm_planes_bench.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedisentangle_ssl_amd import ops_gemm as og  # noqa: E402
from kbench import timeit  # noqa: E402

dev = torch.device("cuda")


def check(m, hb, k, n, act, bias, init, planes_out):
    g = torch.Generator(device="cuda").manual_seed(m + k)
    if hb:
        z = torch.randn(m, hb, k, device=dev, generator=g) * torch.exp(torch.randn(m, 1, 1, device=dev, generator=g))
        a = z.permute(1, 0, 2)
        w = torch.randn(hb, k, n, device=dev, generator=g) * 0.1
    else:
        a = torch.randn(m, k, device=dev, generator=g) * torch.exp(torch.randn(m, 1, device=dev, generator=g))
        w = torch.randn(k, n, device=dev, generator=g) * 0.1
    H = max(hb, 1)
    b = torch.randn(H * n, device=dev, generator=g) if bias else None
    ini = torch.randn(m, H * n, device=dev, generator=g) if init else None
    ref = (torch.bmm(a.double(), w.double()).permute(1, 0, 2).reshape(m, H * n) if hb else a.double() @ w.double())
    if b is not None:
        ref = ref + b.double()
    if ini is not None:
        ref = ref + ini.double()
    ref = {0: lambda t: t, 1: torch.nn.functional.elu, 2: lambda t: torch.nn.functional.leaky_relu(t, 0.01)}[act](ref)
    ap = og.split_planes(a)
    back = ap.to_f32()
    e_rt = float((back.double() - a.double()).abs().max() / a.abs().max())
    bound = ref.abs().max().float().reshape(1) * 1.01 if planes_out else None
    out, pl = og.linear_planes(ap, og.presplit_rm(w), n, b, ini, act, 0.01, True, bound)
    e = float((out.double() - ref).abs().max() / ref.abs().max())
    e_old = float((og._forward(a, w, b, ini, act, 0.01).double() - ref).abs().max() / ref.abs().max())
    msg = f"M={m} hb={hb} K={k} N={n} act={act} bias={bias} init={init}: err {e:.2e} (fp32-input kernel {e_old:.2e}, planes round trip {e_rt:.1e})"
    if pl is not None:
        ep = float((pl.to_f32().double() - ref).abs().max() / ref.abs().max())
        msg += f" plane-out err {ep:.2e}"
        e = max(e, ep)
    print(msg + ("" if e < 1e-6 else "   <<<<<< FAIL"), flush=True)
    return e < 1e-6


def bench(M):
    res = []
    # projection: Z planes [M,8,256] x [8,256,256] -> ELU -> head planes (+ fp32)
    z = torch.randn(M, 8, 256, device=dev)
    a = z.permute(1, 0, 2)
    w = torch.randn(8, 256, 256, device=dev) * 0.05
    am = og.amax(a)
    ws, wr = og.presplit(w), og.presplit_rm(w)
    ap = og.split_planes(a, am)
    bound = (am * w.abs().sum(1).max()).reshape(1)
    res.append(("proj fp32-in -> fp32", timeit(lambda: og._forward(a, w, None, None, 1, 0.0, am, ws), 7)))
    res.append(("proj planes -> planes", timeit(lambda: og.linear_planes(ap, wr, 256, None, None, 1, 0.0, False, bound), 7)))
    res.append(("proj planes -> fp32", timeit(lambda: og.linear_planes(ap, wr, 256, None, None, 1, 0.0, True, None), 7)))
    res.append(("proj planes -> both", timeit(lambda: og.linear_planes(ap, wr, 256, None, None, 1, 0.0, True, bound), 7)))
    del z, a, ap
    # fuser: [M,2048] x [2048,256] + bias, leaky
    h = torch.randn(M, 2048, device=dev)
    w = torch.randn(2048, 256, device=dev) * 0.02
    b = torch.randn(256, device=dev)
    am = og.amax(h)
    ws, wr = og.presplit(w), og.presplit_rm(w)
    hp = og.split_planes(h, am)
    bound = (am * w.abs().sum(0).max() + 4).reshape(1)
    res.append(("fuser fp32-in -> fp32", timeit(lambda: og._forward(h, w, b, None, 2, 0.01, am, ws), 7)))
    res.append(("fuser planes -> fp32", timeit(lambda: og.linear_planes(hp, wr, 256, b, None, 2, 0.01, True, None), 7)))
    res.append(("fuser planes -> both", timeit(lambda: og.linear_planes(hp, wr, 256, b, None, 2, 0.01, True, bound), 7)))
    del h, hp
    # P / Q: x [M,256] x [256,2048]
    x = torch.randn(M, 256, device=dev)
    w = torch.randn(256, 2048, device=dev) * 0.05
    am = og.amax(x)
    ws, wr = og.presplit(w), og.presplit_rm(w)
    xp = og.split_planes(x, am)
    res.append(("P/Q fp32-in -> fp32", timeit(lambda: og._forward(x, w, None, None, 0, 0.0, am, ws), 7)))
    res.append(("P/Q planes -> fp32", timeit(lambda: og.linear_planes(xp, wr, 2048, None, None, 0, 0.0, True, None), 7)))
    res.append(("split_planes x [M,256]", timeit(lambda: og.split_planes(x, am), 7)))
    for k, v in res:
        print(f"{k:28s} {v:7.3f} ms", flush=True)


if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ok = True
    for args in [(1000, 0, 256, 256, 0, False, False, False), (70001, 8, 256, 256, 1, False, False, True),
                 (3333, 0, 2048, 256, 2, True, False, True), (513, 0, 64, 512, 0, False, True, False),
                 (40000, 4, 128, 256, 1, True, True, True), (128, 0, 96, 256, 0, False, False, False),
                 (200000, 0, 256, 2048, 0, False, False, False)]:
        ok &= check(*args)
    print("ALL OK" if ok else "FAILURES", flush=True)
    if M > 0:
        bench(M)
