"""Downstream node-classification trainer on top of the DISGAT path (mirror of the reference's
trainer.py:16-32, 150-223, 297-320).  A caller of the hot path, kept for drop-in completeness."""
import torch
import torch.nn.functional as F

from . import ops, parallel
from .graph import graph_of
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
        self.full_metrics = False
        tr, va, te, self.class_num_mat = split(labels.cpu(), train_ratio=args.node_sup_ratio)
        self.idx_train, self.idx_val, self.idx_test = tr.to(dev), va.to(dev), te.to(dev)

    def get_em(self, feature, adj):
        return fuse_feature(self.models[0].get_em(feature, adj, [self.fuse1, self.fuse2]), fuse=self.args.fuse)

    def reg_fuser(self):
        # one concatenation + one reduction instead of an abs / sum / add chain per parameter (a launch-bound step on small
        # graphs spends ~50 launches there, forward and backward)
        flat = torch.cat([p.reshape(-1) for m in (self.fuse1, self.fuse2) for p in m.parameters()])
        return self.args.reg_weight * flat.abs().sum()

    def _local(self, idx, graph):
        """Node ids of `idx` (global) that this process owns, as local row ids, plus the global count."""
        n_glob = int(idx.shape[0])
        if isinstance(graph, parallel.DistGraph) and graph.world > 1:
            lo = graph.row_start
            keep = (idx >= lo) & (idx < lo + graph.n)
            return idx[keep] - lo, idx[keep], n_glob
        return idx, idx, n_glob

    def _nll_acc(self, output, labels, idx, graph):
        """F.nll_loss(output[idx], labels[idx]) and utils.accuracy over the GLOBAL index set (trainer.py:190-191): on a
        row shard every rank sums over the nodes it owns and the sums are all-reduced; the loss value is global, its
        gradient this rank's share (parameter gradients are summed over ranks afterwards)."""
        loc, glob, n_glob = self._local(idx, graph)
        lab = labels[glob]
        logp = output[loc]
        nll_sum = -logp.gather(1, lab.unsqueeze(1)).sum()
        correct = (logp.argmax(1) == lab).sum().to(torch.float64)
        if isinstance(graph, parallel.DistGraph) and graph.world > 1:
            local = nll_sum / n_glob
            tot = torch.stack([local.detach().to(torch.float64), correct])
            parallel.all_reduce_sum(tot, graph)
            return local + (tot[0].to(local.dtype) - local.detach()), tot[1] / n_glob
        return nll_sum / n_glob, correct / n_glob

    def _row_codes(self, labels, graph):
        """int32 code per row this process owns for ops.cls_loss: label of a training node, label + 65536 of a validation
        node, -1 otherwise.  Built once per (labels, graph): the splits never change (trainer.py:47)."""
        hit = self.__dict__.get("_cls_codes")
        if hit is not None and hit[0] is labels and hit[1] is graph and hit[2] == labels._version:
            return hit[3]
        # once per (labels, graph), outside any captured step: the kernel indexes a row's logits with its label (F.nll_loss
        # raises on an out-of-range target; a label >= 65536 would also spill into the split bit of the code)
        n_cls = int(self.classifier.model[-1].out_features) if hasattr(self.classifier, "model") else 65535
        lo, hi = (int(v) for v in torch.aminmax(labels)) if labels.numel() else (0, 0)
        if lo < 0 or hi >= min(n_cls, 65536):
            raise ValueError(f"class labels must lie in [0, {min(n_cls, 65536)}): got [{lo}, {hi}]")
        code = torch.full((graph.n,), -1, dtype=torch.int32, device=labels.device)
        for split, idx in ((1, self.idx_val), (0, self.idx_train)):
            loc, glob, _ = self._local(idx, graph)
            code[loc] = labels[glob].to(torch.int32) + (split << 16)
        self._cls_codes = (labels, graph, labels._version, code)
        return code

    def _loss_and_logs(self, feature, adj, labels, graph):
        """Forward + the step's four log numbers from ONE pass over the logits (ops.cls_loss: log_softmax, NLL and accuracy
        of the training split, the same on the validation split - trainer.py:186-199, 209-212 read them off the same
        pre-step output).  Returns (loss with graph, acc_train, loss_val, acc_val, log-probabilities)."""
        logits = self.classifier(self.get_em(feature, adj))
        n_tr, n_va = max(1, int(self.idx_train.shape[0])), max(1, int(self.idx_val.shape[0]))
        loss, logp, res = ops.cls_loss(logits, self._row_codes(labels, graph), 0, n_tr, n_va)
        if isinstance(graph, parallel.DistGraph) and graph.world > 1:
            # every rank summed over the nodes it owns: the value is the sum over ranks, the gradient this rank's share
            tot = res.clone()
            parallel.all_reduce_sum(tot, graph)
            return loss + (tot[0].to(loss.dtype) - loss.detach()), tot[1], tot[2], tot[3], logp
        return loss, res[1], res[2], res[3], logp

    def train_step(self, data, labels, epoch):
        """trainer.py:178-223.  Log values are 0-d device tensors (utils.resolve_logs); the sklearn ROC / macro-F1 of
        the validation split, which the reference recomputes on the host every step for its log line only, are
        reported by test() and, when `full_metrics` is set, here too."""
        self._begin_step()
        feature, adj = data
        graph = graph_of(adj)
        sharded = isinstance(graph, parallel.DistGraph) and graph.world > 1
        loss_log, acc_train, loss_val, acc_val, output = self._loss_and_logs(feature, adj, labels, graph)
        reg_log = self.reg_fuser()
        loss = loss_log
        if self.args.reg:          # the L1 term is replicated on every rank: each contributes 1/world of its gradient
            loss = loss_log + (reg_log / graph.world if sharded else reg_log)
        self._finish_step(loss, graph, always_step=True)
        log = {"loss_train": loss_log.detach(), "acc_train": acc_train, "loss_reg": reg_log.detach(),
               "loss_val": loss_val, "acc_val": acc_val}
        if self.full_metrics and not sharded:
            log["roc_val"], log["macroF_val"] = roc_f(output[self.idx_val], labels[self.idx_val])
        return log

    def _static_device(self, adam, data, labels, epoch=0):
        feature, adj = data
        graph = graph_of(adj)
        loss_log, acc_train, loss_val, acc_val, _ = self._loss_and_logs(feature, adj, labels, graph)
        reg_log = self.reg_fuser()
        loss = loss_log + reg_log if self.args.reg else loss_log
        self._static_finish(adam, loss, always_step=True)
        return {"loss_train": loss_log.detach(), "acc_train": acc_train, "loss_reg": reg_log.detach(),
                "loss_val": loss_val, "acc_val": acc_val}

    def test(self, data, labels, epoch=0):
        for m in self.models:
            m.eval()
        feature, adj = data
        graph = graph_of(adj)
        with torch.no_grad():
            output = self.classifier(self.get_em(feature, adj), cls=True)
            loss, acc = self._nll_acc(output, labels, self.idx_test, graph)
            if isinstance(graph, parallel.DistGraph) and graph.world > 1:
                output = parallel.all_gather_rows(output, graph)      # the host metrics need every node's prediction
        roc, f1 = roc_f(output[self.idx_test], labels[self.idx_test])
        return {"loss_test": loss.item(), "acc_test": acc.item(), "roc_test": roc, "macroF_test": f1}
