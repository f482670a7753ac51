"""Downstream node-classification trainer on top of the DISGAT path (mirror of the reference's
trainer.py:16-32, 150-223, 297-320).  A caller of the hot path, kept for drop-in completeness."""
import torch
import torch.nn.functional as F

from .models import MLP
from .pretrainer import Trainer, make_adam
from .utils import accuracy, split  # noqa: F401  (split re-exported: callers import it from here too)


def fuse_feature(feature_list, fuse="last"):
    if fuse == "last":
        return feature_list[-1]
    if fuse == "avg":
        return torch.mean(torch.stack(feature_list))            # sic: trainer.py:21 averages everything
    return torch.cat(feature_list, dim=-1)


def cal_feat_dim(args):
    return args.nhid * (args.enc_layer if args.fuse == "concat" else 1)


def roc_f(output, labels):
    from sklearn.metrics import f1_score, roc_auc_score
    lab = labels.detach().cpu()
    prob = F.softmax(output, dim=-1).detach().cpu()
    try:
        auc = roc_auc_score(lab, prob, average="macro", multi_class="ovr") if labels.max() > 1 else \
            roc_auc_score(lab, prob[:, 1], average="macro")
    except ValueError:
        auc = float("nan")
    return auc, f1_score(lab, torch.argmax(output, dim=-1).detach().cpu(), average="macro")


class ClsTrainer(Trainer):
    def __init__(self, args, model, labels, weight=1.0):
        super().__init__(args, model, weight)
        dev = next(model.parameters()).device
        self.in_dim = cal_feat_dim(args)
        self.classifier = MLP(in_feat=self.in_dim, hidden_size=args.nhid, out_size=labels.max().item() + 1,
                              layers=args.cls_layer).to(dev)
        self.models.append(self.classifier)
        self.models_opt.append(make_adam(self.classifier.parameters(), args))
        tr, va, te, self.class_num_mat = split(labels.cpu(), train_ratio=args.node_sup_ratio)
        self.idx_train, self.idx_val, self.idx_test = tr.to(dev), va.to(dev), te.to(dev)

    def get_em(self, feature, adj):
        return fuse_feature(self.models[0].get_em(feature, adj, [self.fuse1, self.fuse2]), fuse=self.args.fuse)

    def reg_fuser(self):
        l1 = sum(p.abs().sum() for p in self.fuse1.parameters()) + sum(p.abs().sum() for p in self.fuse2.parameters())
        return self.args.reg_weight * l1

    def train_step(self, data, labels, epoch):
        self._begin_step()
        feature, adj = data
        output = self.classifier(self.get_em(feature, adj), cls=True)
        loss_log = F.nll_loss(output[self.idx_train], labels[self.idx_train])
        acc_train = accuracy(output[self.idx_train], labels[self.idx_train])
        reg_log = self.reg_fuser()
        loss = loss_log + reg_log if self.args.reg else loss_log
        (loss * self.loss_weight).backward()
        for opt in self.models_opt:
            opt.step()
        with torch.no_grad():
            loss_val = F.nll_loss(output[self.idx_val], labels[self.idx_val])
            acc_val = accuracy(output[self.idx_val], labels[self.idx_val])
        roc_val, f_val = roc_f(output[self.idx_val], labels[self.idx_val])
        return {"loss_train": loss_log.item(), "acc_train": acc_train.item(), "loss_reg": reg_log.item(),
                "loss_val": loss_val.item(), "acc_val": acc_val.item(), "roc_val": roc_val, "macroF_val": f_val}

    def test(self, data, labels, epoch=0):
        for m in self.models:
            m.eval()
        feature, adj = data
        with torch.no_grad():
            output = self.classifier(self.get_em(feature, adj), cls=True)
            loss = F.nll_loss(output[self.idx_test], labels[self.idx_test])
            acc = accuracy(output[self.idx_test], labels[self.idx_test])
        roc, f1 = roc_f(output[self.idx_test], labels[self.idx_test])
        return {"loss_test": loss.item(), "acc_test": acc.item(), "roc_test": roc, "macroF_test": f1}
