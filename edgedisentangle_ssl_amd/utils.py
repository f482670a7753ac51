"""CLI flags and small loss helpers on the DISGAT path (mirrors /root/reference/utils.py).

get_parser() keeps the reference's flag names, types and defaults (utils.py:23-110) so that
command lines written for the reference's main.py parse unchanged.  Only --model=DISGAT is
implemented by this package; the other encoders are out of scope.
"""
import argparse
import random

import numpy as np
import torch

_FLAGS = [
    # (name, kwargs)                                                           reference line
    ("--no-cuda", dict(action="store_true", default=False)),                  # utils.py:25
    ("--sparse", dict(action="store_true", default=False)),                   # :27
    ("--seed", dict(type=int, default=4)),
    ("--nhid", dict(type=int, default=64)),
    ("--nclass", dict(type=int, default=5)),
    ("--dataset", dict(type=str, default="dblp")),
    ("--size", dict(type=int, default=64)),
    ("--epochs", dict(type=int, default=510)),
    ("--lr", dict(type=float, default=0.01)),
    ("--weight_decay", dict(type=float, default=5e-4)),
    ("--dropout", dict(type=float, default=0.1)),
    ("--batch_nums", dict(type=int, default=6000)),
    ("--load", dict(type=int, default=None)),
    ("--save", dict(type=str, default=None)),
    ("--log", dict(action="store_true", default=False)),
    ("--method", dict(type=str, default="no", choices=["no"])),
    ("--model", dict(type=str, default="DISGAT",
                     choices=["sage", "gcn", "GAT", "sage2", "MLP", "RGCN", "HAN", "DISGAT", "GIN", "FactorGCN",
                              "Mixhop", "H2GCN"])),
    ("--nhead", dict(type=int, default=4)),
    ("--hetero", dict(action="store_true", default=False)),
    ("--hnn", dict(action="store_true", default=False)),
    ("--edge_num", dict(type=int, default=3)),
    ("--used_edge", dict(type=int, default=1)),
    ("--cls_layer", dict(type=int, default=2)),
    ("--EdgePred_layer", dict(type=int, default=1)),
    ("--downstream", dict(nargs="+", type=str, choices=["CLS", "Edge"])),
    ("--down_weight", dict(nargs="+", type=float)),
    ("--pretrain", dict(nargs="+", type=str,
                        choices=["PredAttr", "PredDistance", "PredContext", "DisEdge", "SupEdge", "DifHead"])),
    ("--pre_weight", dict(nargs="+", type=float)),
    ("--pre_edge", dict(nargs="+", type=int)),
    ("--finetune", dict(action="store_true", default=False)),
    ("--enc_layer", dict(type=int, default=2)),
    ("--fuse", dict(type=str, default="last", choices=["last", "avg", "concat"])),
    ("--pretext_dim", dict(type=int, default=16)),
    ("--cluster_num", dict(type=int, default=16)),
    ("--node_sup_ratio", dict(type=float, default=0.25)),
    ("--reg", dict(action="store_true", default=False)),
    ("--reg_weight", dict(type=float, default=0.01)),
    ("--batch", dict(action="store_true", default=False)),
    ("--batch_size", dict(type=int, default=40)),
    ("--SubgraphSize", dict(type=int, default=128)),
    ("--origin_feat", dict(action="store_true", default=False)),
    ("--att", dict(type=int, default=2)),                                     # :92
    ("--dis_type", dict(type=int, default=1)),
    ("--constrain_layer", dict(type=int, default=0)),
    ("--residue", dict(action="store_true", default=False)),
    ("--fuse_no_relu", dict(action="store_true", default=False)),
    ("--residue_type", dict(type=int, default=0)),
    ("--steps", dict(type=int, default=5)),
    ("--gnn_type", dict(type=str, default="AT", choices=["AT", "SAGE", "GCN"])),   # :103
    ("--case", dict(action="store_true", default=False)),
    ("--conformT", dict(action="store_true", default=False)),
]


def get_parser():
    parser = argparse.ArgumentParser()
    for name, kw in _FLAGS:
        parser.add_argument(name, **kw)
    return parser


def adj_mse_loss(adj_rec, adj_tgt):
    """utils.py:287-298, including its quirk: `total` is shape[0]**2, so for the 1-D [M] tensors the
    sparse SSL losses pass in, negatives are weighted by n_pos / (M^2 - n_pos)."""
    edge_num = (adj_tgt != 0).sum()
    total_num = adj_tgt.shape[0] ** 2
    neg_weight = edge_num / (total_num - edge_num)
    weight = torch.where(adj_tgt == 0, neg_weight.to(adj_rec.dtype), torch.ones((), dtype=adj_rec.dtype, device=adj_rec.device))
    return torch.mean(weight * (adj_rec - adj_tgt) ** 2)


def accuracy(output, labels):
    preds = output.max(1)[1].type_as(labels)
    return preds.eq(labels).double().sum() / len(labels)


def group_correlation(embedding):
    """Pearson correlation between the rows of `embedding` [R, C] -> [R, R] (utils.py:326-334: each row centred by
    its own mean, Gram matrix divided by the outer product of the row norms)."""
    centred = embedding - embedding.mean(dim=-1, keepdim=True)
    gram = centred @ centred.t()
    norm = torch.sqrt(torch.diagonal(gram))
    return gram / torch.outer(norm, norm)


def split(labels, train_ratio=0.25):
    """Per-class random train/val/test split (utils.py:118-161); same draw order from `random`."""
    val_ratio, test_ratio = (1 - train_ratio) / 4, (1 - train_ratio) / 4 * 3
    num_classes = len(set(labels.tolist()))
    train_idx, val_idx, test_idx = [], [], []
    c_num_mat = np.zeros((num_classes, 3)).astype(int)
    for i in range(num_classes):
        c_idx = (labels == i).nonzero()[:, -1].tolist()
        random.shuffle(c_idx)
        if len(c_idx) < 11:
            raise ValueError("too small class type: {}, num{}".format(i, len(c_idx)))
        c_num_mat[i] = [int(len(c_idx) * train_ratio), int(len(c_idx) * val_ratio), int(len(c_idx) * test_ratio)]
        a, b, c = c_num_mat[i]
        train_idx += c_idx[:a]
        val_idx += c_idx[a:a + b]
        test_idx += c_idx[a + b:a + b + c]
    random.shuffle(train_idx)
    return torch.LongTensor(train_idx), torch.LongTensor(val_idx), torch.LongTensor(test_idx), c_num_mat


def resolve_logs(log):
    """Dict with 0-d device tensors among its values -> plain Python numbers, ONE device-to-host transfer."""
    keys = [k for k, v in log.items() if torch.is_tensor(v)]
    if keys:
        vals = torch.stack([log[k].detach().to(torch.float64).reshape(()) for k in keys]).tolist()
        log = dict(log)
        log.update(zip(keys, vals))
    return log
