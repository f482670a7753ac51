"""End-to-end drop-in run on the GPU: the reference's main.py flow (CLS fine-tuning + SupEdge +
DisEdge + DifHead train_steps, dropout 0.1, Adam) through the fused HIP forward/backward."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gnn,att", [("AT", 3), ("SAGE", 1), ("GCN", 2)])
def test_main_flow_trains(golden_dir, gnn, att):
    from edgedisentangle_ssl_amd import main
    argv = ["--model=DISGAT", "--sparse", "--dataset", "chameleon", "--fixture", os.path.join(golden_dir, "data_chameleon.npz"),
            "--gnn_type", gnn, "--att", str(att), "--nhead", "4", "--nhid", "32", "--epochs", "8", "--steps", "2",
            "--downstream", "CLS", "--down_weight", "1.0", "--finetune", "--pretrain", "SupEdge", "DisEdge", "DifHead",
            "--pre_weight", "1", "1", "1", "--pre_edge", "1", "1", "1", "--dropout", "0.1", "--seed", "4", "--quiet"]
    hist = main.run(argv)
    assert len(hist) == 8
    for h in hist:
        for k in ("loss_train", "loss_heads_sup", "loss_head_disen", "loss_head_diversity"):
            assert math.isfinite(h[k]), (k, h)
    assert "test_acc_test" in hist[0]
    # the optimisers really step through the fused backward: the training losses move down
    assert hist[-1]["loss_head_diversity"] < hist[0]["loss_head_diversity"]
    assert hist[-1]["loss_train"] < hist[0]["loss_train"]
