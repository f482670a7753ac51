#!/usr/bin/env python3
"""torch.profiler view (ATen op x input shapes) of one C4 training iteration: finds the host-side ops around the kernels."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
o = bench.parse()
dev = torch.device("cuda")
args, enc, (sup, dis, dif), graph, x, lists = bench.build_workload(o, 0, 1, dev)
for m in list(enc.modules()):
    if hasattr(m, "dropout"):
        m.dropout = 0.1
(si, sl), (hi, hl), (ti, tl) = lists
data = (x, graph)


def train():
    for tr, fn in ((sup, lambda: sup.loss(data, sl, [si])), (dis, lambda: dis.loss(data, [hl, tl], [hi, ti])),
                   (dif, lambda: dif.loss(data))):
        tr._begin_step()
        loss = fn()
        tr._finish_step(loss, graph)


train()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=40,
                                                         max_shapes_column_width=70))
