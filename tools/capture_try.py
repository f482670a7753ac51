#!/usr/bin/env python3
"""Epoch time of the reference's training flow on a bundled graph, eager vs captured (main.run --capture off|on)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgedisentangle_ssl_amd import main as drop_in  # noqa: E402

names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["chameleon", "cora"]
modes = ("on",) if "--on" in sys.argv else ("off",) if "--off" in sys.argv else ("off", "on")
for name in names:
    fx = os.path.join(ROOT, "tests", "golden", f"data_{name}.npz")
    argv = ["--model=DISGAT", "--sparse", "--dataset", name, "--fixture", fx, "--gnn_type", "AT", "--att", "3",
            "--nhead", "8", "--nhid", "64", "--steps", "5", "--downstream", "CLS", "--down_weight", "1.0", "--finetune",
            "--pretrain", "SupEdge", "DisEdge", "DifHead", "--pre_weight", "1", "1", "1", "--pre_edge", "1", "1", "1",
            "--dropout", "0.1", "--seed", "4", "--quiet"]
    for mode in modes:
        drop_in.run(argv + ["--epochs", "2", "--capture", mode])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hist = drop_in.run(argv + ["--epochs", "12", "--capture", mode])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 12 * 1e3
        last = {k: round(v, 5) for k, v in hist[-1].items() if k.startswith("loss") or k.startswith("acc")}
        t0 = time.perf_counter()
        drop_in.run(argv + ["--epochs", "60", "--capture", mode])
        torch.cuda.synchronize()
        t60 = (time.perf_counter() - t0) * 1e3
        print(f"{name:10s} capture={mode:3s} {dt:8.2f} ms/epoch over 12 epochs incl. set-up and capture; steady state "
              f"{(t60 - dt * 12) / 48:7.2f} ms/epoch (60-epoch run minus 12-epoch run); last epoch: {last}", flush=True)
