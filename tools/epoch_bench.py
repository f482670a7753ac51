#!/usr/bin/env python3
"""Wall time per epoch of the reference's training flow (CLS fine-tuning x --steps + SupEdge + DisEdge +
DifHead train_steps, dropout 0.1, Adam) on the bundled small graphs (BASELINE configs[1])."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgedisentangle_ssl_amd import main  # noqa: E402

for name in ("chameleon", "cora", "cora_full"):
    argv = ["--model=DISGAT", "--sparse", "--dataset", name, "--fixture", os.path.join(ROOT, "tests/golden", f"data_{name}.npz"),
            "--gnn_type", "AT", "--att", "3", "--nhead", "8", "--nhid", "64", "--steps", "5", "--downstream", "CLS",
            "--down_weight", "1.0", "--finetune", "--pretrain", "SupEdge", "DisEdge", "DifHead", "--pre_weight", "1", "1", "1",
            "--pre_edge", "1", "1", "1", "--dropout", "0.1", "--seed", "4", "--quiet"]
    main.run(argv + ["--epochs", "3"])                 # warm-up (library load, hipBLASLt heuristics)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 10
    hist = main.run(argv + ["--epochs", str(n)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"{name:10s} {dt * 1e3:8.1f} ms / epoch (5 CLS steps + 3 SSL steps)   final losses: "
          f"cls {hist[-1]['loss_train']:.4f} sup {hist[-1]['loss_heads_sup']:.5f} dis {hist[-1]['loss_head_disen']:.5f} "
          f"dif {hist[-1]['loss_head_diversity']:.4f}")
