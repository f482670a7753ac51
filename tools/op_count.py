#!/usr/bin/env python3
"""ATen / library launch counts of one static (capturable) train_step per trainer on a bundled graph: where a launch-bound
epoch's kernels come from."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if args else "chameleon"
att = sys.argv[sys.argv.index("--att") + 1] if "--att" in sys.argv else "3"        # tools/op_count.py cora --att 2
if "--att" in sys.argv:
    args = [a for a in args if a != att]
    name = args[0] if args else "chameleon"
from edgedisentangle_ssl_amd import main as drop_in, pretrainer, trainer  # noqa: E402

# build the trainers exactly as main.run does, by running two eager epochs through it with hooks that keep the objects
kept = {}
for cls in (trainer.ClsTrainer, pretrainer.SupEdgeTrainer, pretrainer.GeneratedEdgeTrainer, pretrainer.DifHeadTrainer):
    orig = cls.train_step_captured

    def wrap(self, *a, _o=orig, _n=cls.__name__):
        kept[_n] = (self, a)
        return _o(self, *a)
    cls.train_step_captured = wrap
fx = os.path.join(ROOT, "tests", "golden", f"data_{name}.npz")
argv = ["--model=DISGAT", "--sparse", "--dataset", name, "--fixture", fx, "--gnn_type", "AT", "--att", att, "--nhead", "8", "--nhid", "64",
        "--steps", "1", "--downstream", "CLS", "--down_weight", "1.0", "--finetune", "--pretrain", "SupEdge", "DisEdge", "DifHead",
        "--pre_weight", "1", "1", "1", "--pre_edge", "1", "1", "1", "--dropout", "0.1", "--seed", "4", "--quiet", "--epochs", "2", "--capture", "on"]
drop_in.run(argv)
from torch.profiler import ProfilerActivity, profile  # noqa: E402
for n, (tr, a) in kept.items():
    st = tr.static_step()
    st.run_eager(*a)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        st.run_eager(*a)
        torch.cuda.synchronize()
    ev = prof.key_averages()
    kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
    tot = sum(e.count for e in kern)
    print(f"\n== {n}: {tot} device launches per step")
    for e in sorted(kern, key=lambda e: -e.count)[:int(os.environ.get('OP_COUNT_TOP', 14))]:
        print(f"   {e.count:4d}  {e.key[:110]}")
    ops = [e for e in ev if e.device_type != torch.autograd.DeviceType.CUDA and e.key.startswith("aten::")]
    print("   aten ops:", ", ".join(f"{e.key[6:]} x{e.count}" for e in sorted(ops, key=lambda e: -e.count)[:28]))
