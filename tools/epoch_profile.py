#!/usr/bin/env python3
"""cProfile of the small-graph training epoch (host side): tools/epoch_profile.py [dataset]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgedisentangle_ssl_amd import main as drop_in  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "chameleon"
fx = os.path.join(ROOT, "tests", "golden", f"data_{name}.npz")
argv = ["--model=DISGAT", "--sparse", "--dataset", name, "--fixture", fx, "--gnn_type", "AT", "--att", "3", "--nhead", "8",
        "--nhid", "64", "--steps", "5", "--downstream", "CLS", "--down_weight", "1.0", "--finetune", "--pretrain", "SupEdge",
        "DisEdge", "DifHead", "--pre_weight", "1", "1", "1", "--pre_edge", "1", "1", "1", "--dropout", "0.1", "--seed", "4", "--quiet"]
drop_in.run(argv + ["--epochs", "3"])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
drop_in.run(argv + ["--epochs", "10"])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
