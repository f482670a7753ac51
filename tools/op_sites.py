#!/usr/bin/env python3
"""Which lines of the package launch a static train_step's device work: per trainer, the ATen ops / C-ABI calls that
reach the GPU, grouped by the innermost edgedisentangle_ssl_amd frame that issued them (tools/op_count.py says how
many launches there are; this says where they come from)."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
name = sys.argv[1] if len(sys.argv) > 1 else "chameleon"
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
from edgedisentangle_ssl_amd import main as drop_in, pretrainer, trainer  # noqa: E402

kept = {}
for cls in (trainer.ClsTrainer, pretrainer.SupEdgeTrainer, pretrainer.GeneratedEdgeTrainer, pretrainer.DifHeadTrainer):
    orig = cls.train_step_captured

    def wrap(self, *a, _o=orig, _n=cls.__name__):
        kept[_n] = (self, a)
        return _o(self, *a)
    cls.train_step_captured = wrap
fx = os.path.join(ROOT, "tests", "golden", f"data_{name}.npz")
argv = ["--model=DISGAT", "--sparse", "--dataset", name, "--fixture", fx, "--gnn_type", "AT", "--att", "3", "--nhead", "8", "--nhid", "64",
        "--steps", "1", "--downstream", "CLS", "--down_weight", "1.0", "--finetune", "--pretrain", "SupEdge", "DisEdge", "DifHead",
        "--pre_weight", "1", "1", "1", "--pre_edge", "1", "1", "1", "--dropout", "0.1", "--seed", "4", "--quiet", "--epochs", "2", "--capture", "on"]
drop_in.run(argv)
from torch.profiler import ProfilerActivity, profile  # noqa: E402

PKG = os.sep + "edgedisentangle_ssl_amd" + os.sep


def site_of(stack):
    for fr in stack:                       # innermost first
        if PKG in fr and "tools" + os.sep not in fr:
            return fr.split(PKG)[-1].strip()
    return "(outside the package: autograd engine / optimizer)"


for n, (tr, a) in kept.items():
    if only and n not in only:
        continue
    st = tr.static_step()
    st.run_eager(*a)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        st.run_eager(*a)
        torch.cuda.synchronize()
    # launches per CPU op: a device event is linked to the runtime call that made it, whose parent chain holds the op
    by_site = collections.Counter()
    by_site_ops = collections.defaultdict(collections.Counter)
    n_launch = 0
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            continue
        k = sum(1 for _ in ev.kernels) if ev.kernels else 0
        if not k:
            continue
        # count only at the outermost op that still has a package frame right above it: skip children of counted ops
        p = ev.cpu_parent
        nested = False
        while p is not None:
            if p.kernels:
                nested = True
                break
            p = p.cpu_parent
        if nested:
            continue
        n_launch += k
        s = site_of(ev.stack or [])
        by_site[s] += k
        by_site_ops[s][ev.name] += k
    if os.environ.get("OP_SITES_SEQ") == "1":       # the step as a time-ordered list: op, launches, enclosing ops
        print(f"\n== {n}: sequence")
        for ev in sorted(prof.events(), key=lambda e: e.time_range.start):
            if ev.device_type == torch.autograd.DeviceType.CUDA or not ev.kernels:
                continue
            if any(c.kernels for c in ev.cpu_children):
                continue                                  # report at the innermost op that owns the launches
            chain, p = [], ev.cpu_parent
            while p is not None:
                chain.append(p.name.replace("aten::", ""))
                p = p.cpu_parent
            kn = ",".join(k.name.split("(")[0].split("::")[-1][:28] for k in ev.kernels)
            print(f"   {len(ev.kernels):2d} {ev.name.replace('aten::', ''):28s} <- {' <- '.join(chain[:4]):60s} [{kn}]")
    if os.environ.get("OP_SITES_KERNELS") == "1":   # every device launch in time order, with the CPU op that made it (if any)
        owner = {}
        for ev in prof.events():
            if ev.device_type != torch.autograd.DeviceType.CUDA and ev.kernels and not any(c.kernels for c in ev.cpu_children):
                chain, p = [ev.name.replace("aten::", "")], ev.cpu_parent
                while p is not None and len(chain) < 3:
                    chain.append(p.name.replace("aten::", "").replace("autograd::engine::evaluate_function: ", ""))
                    p = p.cpu_parent
                for k in ev.kernels:
                    owner[(k.name, k.time_range.start if hasattr(k, "time_range") else 0)] = " <- ".join(chain)
        devs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
        print(f"\n== {n}: {len(devs)} device launches in order")
        for e in devs:
            nm = e.name.split("(")[0].replace("void ", "").replace("at::native::", "").replace("disgat::", "D:")[:70]
            print(f"   {nm:70s} {e.time_range.elapsed_us():7.1f} us")
    print(f"\n== {n}: {n_launch} launches attributed")
    for s, c in by_site.most_common(int(os.environ.get("OP_SITES_TOP", 60))):
        ops = ", ".join(f"{o.replace('aten::', '')} x{k}" for o, k in by_site_ops[s].most_common(6))
        print(f"   {c:4d}  {s[:90]:90s} {ops}")
