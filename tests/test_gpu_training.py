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
    if gnn == "AT":
        argv.append("--case")            # main.py:289-301: the disentanglement study every 40th epoch
    hist = main.run(argv)
    assert len(hist) == 8
    if gnn == "AT":
        assert 0.0 < hist[0]["att_correlation_layer1"] <= 1.0 + 1e-6 and 0.0 < hist[0]["att_correlation_layer2"] <= 1.0 + 1e-6
    for h in hist:
        for k in ("loss_train", "loss_heads_sup", "loss_head_disen", "loss_head_diversity"):
            assert math.isfinite(h[k]), (k, h)
    assert "test_acc_test" in hist[0]
    # the optimisers really step through the fused backward: the training losses move down
    assert hist[-1]["loss_head_diversity"] < hist[0]["loss_head_diversity"]
    assert hist[-1]["loss_train"] < hist[0]["loss_train"]


def test_analyze_disentangle_returns_correlation_grids(golden_dir):
    """Trainer.analyze_disentangle (trainer.py:82-134) on the HIP path: per layer a head x head score correlation,
    its mean absolute value and a feature-dimension correlation of the layer output."""
    import inputs_common as ic
    from test_gpu_parity import build, real_inputs
    from edgedisentangle_ssl_amd import pretrainer
    dev = torch.device("cuda")
    x, adj, n, ei, sup, ho, he = real_inputs(golden_dir, "chameleon", dev)
    a, enc, fus = build("AT", 3, 4, 32, x.shape[1], 321, dev)
    a.lr, a.weight_decay, a.dis_type = 0.01, 5e-4, 1
    tr = pretrainer.SupEdgeTrainer(a, enc, 1.0)
    ic.load_params(tr.fuse1, 7)
    ic.load_params(tr.fuse2, 8)
    for m in tr.models:
        m.to(dev).eval()
    dist, at_cor, feat_cor = tr.analyze_disentangle(x, adj)
    assert len(dist) == len(at_cor) == len(feat_cor) == 2
    for layer in range(2):
        c = at_cor[layer]
        assert c.shape == (4, 4) and torch.allclose(c, c.t(), atol=1e-5)
        assert torch.allclose(torch.diagonal(c), torch.ones(4, device=dev), atol=1e-4)
        assert 0.0 < dist[layer] <= 1.0 + 1e-6
        assert feat_cor[layer].shape == (32, 32) and torch.isfinite(feat_cor[layer]).all()
