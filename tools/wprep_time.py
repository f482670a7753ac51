#!/usr/bin/env python3
"""Device time of the weight preparation (disgat_split_f16) on the weight shapes of a bundled-graph train_step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgedisentangle_ssl_amd import ops_gemm  # noqa: E402

dev = torch.device("cuda")
base = torch.randn(128, 1024, device=dev)
wide = torch.randn(192, 576, device=dev)
cases = {
    "dense [64,1024]": base[:64],
    "dense [64,1024] second half of a stack": base[64:],
    "transposed [1024,64]": base[:64].t(),
    "[8,64,64] heads": torch.randn(8, 64, 64, device=dev),
    "[8,64,64] heads transposed": torch.randn(8, 64, 64, device=dev).transpose(1, 2),
    "column slice [64,512] of [192,576] transposed": wide[:64, 64:].t(),
    "expanded over heads [8,64,192]": wide[:, :64].t().unsqueeze(0).expand(8, 64, 192),
    "[64,64]": torch.randn(64, 64, device=dev),
    "[512,64]": torch.randn(512, 64, device=dev),
    "large [256,2048] (two-launch form)": torch.randn(256, 2048, device=dev),
}
for name, w in cases.items():
    for _ in range(3):
        ops_gemm.split_weight_f16(w)
    torch.cuda.synchronize()
    n = 50
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):          # device time: the calls replayed from a HIP graph, no host in the loop
            for _ in range(n):
                keep = ops_gemm.split_weight_f16(w)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(4):
            g.replay()
        e1.record(st)
        torch.cuda.synchronize()
    print(f"{name:52s} {e0.elapsed_time(e1) / (4 * n) * 1e3:8.1f} us per call, replayed from a HIP graph")
