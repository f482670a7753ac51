"""SupEdge / DisEdge / DifHead self-supervised trainers on the DISGAT path.

Mirrors of /root/reference/pretrainer.py:658-776 (SupEdgeTrainer), :372-654
(GeneratedEdgeTrainer) and :780-857 (DifHeadTrainer) plus the Trainer base of
trainer.py:35-80: same class names, constructor signature (args, model, weight), attributes
(fuse1/fuse2, models, models_opt), log keys and loss arithmetic.  Differences, all on the input
side of the hot path:
  * the reference samples training pairs through dense N x N masks (pretrainer.py:683-707,
    524-576); here the same distribution (Bernoulli(3*rho) over all entries united with a third
    of the positives, row-major order) is drawn in O(output) work by kernels of the library (sampling.py, csrc/pair_sample.hip);
  * `loss(...)` exposes the forward + loss part of train_step on pre-sampled pairs (what bench.py
    times); train_step = sample + loss + backward + optimiser steps as in the reference;
  * train_step returns its log values as 0-d DEVICE tensors (the reference calls .item() per step): nothing in a step
    waits for the GPU except the sampled lists' lengths (one read per list; none on the captured path); utils.resolve_logs() turns a dict of them into floats with
    one transfer (main.run does that once per epoch).
"""
import torch
import torch.nn.functional as F
from . import ops, ops_bwd, ops_gemm, optim, parallel, sampling
from .graph import graph_of
from .layers import FuseLayer
from .models import MLP
from .utils import adj_mse_loss, split


def make_adam(params, args):
    """The reference's per-module Adam (trainer.py:58-60): same lr / weight_decay / betas / eps and independent state
    per module, but every optimiser of a trainer is stepped by ONE multi-tensor HIP launch (optim.step_all) - a
    trainer owns 3-5 optimisers of a few small tensors each and steps them all every train_step, which on small graphs
    is mostly launch overhead."""
    return optim.ModuleAdam(params, lr=args.lr, weight_decay=args.weight_decay)


def _global_count(t, graph):
    """len(t) summed over the ranks of a sharded graph (python int)."""
    c = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    return int(parallel.all_reduce_sum(c, graph))


def _sampler(graph, pos, n_pos_global=None):
    """The pair sampler of one positive set over this process's rows (sampling.PairSampler: local row ids, global column ids)."""
    return sampling.PairSampler(graph.n, pos, n_cols=graph.n_cols,
                                n_pos_global=_global_count(pos, graph) if n_pos_global is None else n_pos_global)


def _graph_sampler(graph):
    """The sampler whose positives are the graph's own entries (SupEdge, analyze_disentangle): kept with the graph."""
    smp = graph.__dict__.get("_pair_sampler")
    if smp is None:
        smp = graph._pair_sampler = _sampler(graph, sampling.flat_edges(graph))
    return smp


def _pair_loss_value(base, h_lo, h_hi, labels, graph):
    """(loss, neg_w, m, value) of utils.adj_mse_loss on the list.  Unsharded: finished inside the pair-loss launches (value =
    the kernel's {loss, neg_w, m}); on a sharded graph sums and pair count are all-reduced first (global loss; value None)."""
    count = getattr(labels, "_disgat_count", None)      # fixed-capacity list (sampling.PairSampler.sample_static): its valid length, on the device
    if not (isinstance(graph, parallel.DistGraph) and graph.world > 1):
        _acc, loss, value = ops.pair_loss_sums(base, h_lo, h_hi, labels, count=count, want_value=True)
        return loss, value[1], value[2], value
    acc = ops.pair_loss_sums(base, h_lo, h_hi, labels)
    if count is None:
        count = acc.new_full((1,), float(labels.shape[0]))                 # fill kernel: stays graph-capturable
    acc = torch.cat([acc, count.reshape(1).to(acc.dtype)])
    parallel.all_reduce_sum(acc, graph)
    m = acc[3]
    neg_w = acc[2] / (m * m - acc[2])
    return ((acc[0] + neg_w * acc[1]) / m).to(torch.float32), neg_w, m, None


class _PairLoss(torch.autograd.Function):
    """Differentiable weighted-MSE pair loss on the [H,M] score buffer: forward = the partial-sum kernel + its finish (which
    also does the class-weight / mean arithmetic), backward = one kernel writing d loss / d score for every row (the torch
    formulation took ~10 passes over [M] / [H,M] temporaries per loss, and the scalar arithmetic around the kernels another
    15 launches per list).  Sharded: the value is the global loss, the gradient this rank's part (global weights)."""

    @staticmethod
    def forward(ctx, base, labels, h_lo, h_hi, graph):
        loss, neg_w, m, value = _pair_loss_value(base, h_lo, h_hi, labels, graph)
        ctx.fused = value is not None
        if ctx.fused:
            ctx.save_for_backward(base, labels, value)
        else:
            ctx.save_for_backward(base, labels, neg_w, m)
        ctx.rng = (h_lo, h_hi)
        return loss

    @staticmethod
    def backward(ctx, gout):
        h_lo, h_hi = ctx.rng
        from . import _lib
        if ctx.fused:
            base, labels, value = ctx.saved_tensors
            coef = None
            gout = gout.reshape(1).to(torch.float32).contiguous()
        else:
            base, labels, neg_w, m = ctx.saved_tensors
            value = None
            coef = (torch.stack([torch.ones_like(neg_w), neg_w]) * (gout.double() / m)).to(torch.float32).contiguous()
        # [H, M] gradient backed by an [M, H] buffer: the segment passes of the score backward (ops_bwd._g_strides) then find
        # a pair's H values in one 32-byte run - the column-side pass visits the pairs in column order, i.e. at random
        g = torch.empty((base.shape[1], base.shape[0]), dtype=base.dtype, device=base.device)
        _lib.call("disgat_pair_loss_bwd", base.data_ptr(), int(base.shape[1]), int(base.shape[0]), h_lo, h_hi,
                  labels.data_ptr(), ops._ptr(coef), ops._ptr(value), ops._ptr(None if coef is not None else gout), g.data_ptr(), 1,
                  ops._stream())
        return g.t(), None, None, None, None


def pair_mse_loss(aux, h_lo, h_hi, labels, graph=None):
    """pred = sigmoid(sum_{h in [lo,hi)} aux_h); utils.adj_mse_loss(pred, labels) on 1-D input
    (pretrainer.py:734-739, 619-627).  aux: list of H [M,1] tensors (entries outside the range may
    be None).  Without autograd the reduction runs in the HIP pair-loss kernel; on a sharded graph
    the three partial sums and the pair count are all-reduced (the loss is global)."""
    heads = [aux[h] for h in range(h_lo, h_hi)]
    base = heads[0]._base
    sharded = isinstance(graph, parallel.DistGraph) and graph.world > 1
    fused = (base is not None and base.dim() == 2 and base.is_contiguous() and base.shape[1] == labels.shape[0]
             and heads[0].data_ptr() == base[h_lo].data_ptr())
    if fused and labels.dtype == torch.float32 and labels.is_contiguous():
        if torch.is_grad_enabled() and base.requires_grad:
            return _PairLoss.apply(base, labels, h_lo, h_hi, graph)
        return _pair_loss_value(base, h_lo, h_hi, labels, graph)[0]
    pred = torch.sigmoid(torch.sum(torch.stack(heads), dim=0)).squeeze(-1)
    if not sharded:
        return adj_mse_loss(pred, labels)
    # sharded + autograd: class weights and the mean use GLOBAL counts; this rank contributes the
    # weighted squared error of its own pairs (the value returned is the global loss, its gradient
    # the local part - parameter gradients are summed over ranks afterwards)
    cnt = torch.tensor([float((labels != 0).sum()), float(labels.shape[0])], dtype=torch.float64, device=labels.device)
    parallel.all_reduce_sum(cnt, graph)
    neg_w = (cnt[0] / (cnt[1] * cnt[1] - cnt[0])).to(pred.dtype)
    w = torch.where(labels == 0, neg_w, torch.ones((), dtype=pred.dtype, device=pred.device))
    local = (w * (pred - labels) ** 2).sum() / cnt[1].to(pred.dtype)
    return local + (parallel.all_reduce_sum(local.detach().clone(), graph) - local.detach())


class Trainer(object):
    """trainer.py:35-60: owns one fuser pair per trainer and one Adam per sub-module."""

    def __init__(self, args, model, weight):
        self.args = args
        self.loss_weight = weight
        self.models = [model]
        res = (args.size, args.nhid) if args.residue else (0, 0)
        self.fuse1 = FuseLayer(args, args.nhead, nfeat=args.nhid, residue=res[0])
        self.fuse2 = FuseLayer(args, args.nhead, nfeat=args.nhid, residue=res[1])
        dev = next(model.parameters()).device
        self.fuse1.to(dev)
        self.fuse2.to(dev)
        self.models += [self.fuse1, self.fuse2]
        self.models_opt = [make_adam(m.parameters(), args) for m in self.models]

    def _begin_step(self):
        for i, model in enumerate(self.models):
            model.train()
            self.models_opt[i].zero_grad()

    def _finish_step(self, loss, graph=None, always_step=False):
        (loss * self.loss_weight).backward()
        ops_bwd.clear_segment_cache()              # the step's pair lists are not scored again
        parallel.all_reduce_grads(self.models, graph)
        if self.loss_weight != 0 or always_step:   # pretrainer.py:754-756 gates on the weight; trainer.py:205-206 does not
            optim.step_all(self.models_opt)        # every sub-module's Adam in one launch

    def _layer_on(self, i):
        return self.constrain_layer == 0 or self.constrain_layer == i       # pretrainer.py:597, 728

    # ---- the same train_step as a HIP graph (capture.StaticStep): subclasses give the host half (_static_host: the
    # pair samplers' seeds at first use) and the device half (_static_device: sample + forward + loss + backward + Adam).
    def _static_finish(self, adam, loss, always_step=False):
        (loss * self.loss_weight).backward()
        ops_bwd.clear_segment_cache()
        if self.loss_weight != 0 or always_step:
            adam.step()

    def static_step(self):
        from .capture import StaticStep
        st = self.__dict__.get("_static")
        if st is None:
            st = self._static = StaticStep(self, self._static_host, self._static_device)
        return st

    def close(self):
        """Drop the captured step (its HIP graph is destroyed here, by reference count: capture.StaticStep)."""
        st = self.__dict__.pop("_static", None)
        if st is not None:
            st.close()

    def train_step_captured(self, *args):
        """train_step replayed from a HIP graph (unsharded graphs; the arguments must be the same objects every call).
        Returns the step's log dict - 0-d device tensors that the NEXT replay overwrites."""
        st = self.static_step()
        flat = [b for a in args for b in (a if isinstance(a, (list, tuple)) else [a])]
        key = tuple(id(a) for a in flat if not isinstance(a, (int, float, type(None))))
        if st.__dict__.setdefault("_args_key", key) != key:
            raise RuntimeError("train_step_captured: a captured step replays on the tensors it was captured with")
        return st(*args)

    def _static_host(self, *args):
        pass              # nothing: the pair samplers' generator state lives on the device (csrc/pair_sample.hip)

    def samplers(self, *args):
        """The PairSamplers this trainer draws from (main.run reads their event counters)."""
        return []

    def analyze_disentangle(self, feature, adj):
        """trainer.py:82-134: how different are the heads?  Per layer: the head x head correlation of the raw scores
        on a sampled pair list (a third of the edges + 3x as many random pairs), its mean absolute value, and the
        feature-dimension correlation of the layer output.  The reference draws the pairs through dense N x N masks
        and plots the grids; here the pairs come from the O(E) device sampler and the grids are returned."""
        from .utils import group_correlation
        assert self.args.model == "DISGAT", "analyze disentanglement is only implemented for DISGAT"
        fusers = [self.fuse1, self.fuse2]
        g = graph_of(adj)
        with torch.no_grad():
            feats = self.models[0].get_em(feature, adj, fusers)
            indices, _ = _graph_sampler(g).sample()
            scores = self.models[0].predict_adjs_sparse(feature, adj, fusers, auxiliary_edges=[indices])
            at_cor, at_dist, feat_cor = [], [], []
            for layer in range(2):
                feat_cor.append(group_correlation(feats[layer].t()))
                per_head = torch.stack([h[0][:, 0] for h in scores[layer]])            # [H, M]
                cor = group_correlation(per_head)
                at_cor.append(cor)
                at_dist.append(torch.mean(torch.abs(cor)).item())
        return at_dist, at_cor, feat_cor


class SupEdgeTrainer(Trainer):
    """Edge-recovery supervision on the sum of all heads (pretrainer.py:658-776)."""

    def __init__(self, args, model, weight):
        super().__init__(args, model, weight)
        assert args.model == "DISGAT", "supervision on edges can only be conducted on DISGAT model"
        self.sparse = args.sparse
        self.constrain_layer = args.constrain_layer

    def get_label_all(self, feature, adj):
        """The reference returns a dense 0/1 N x N tensor (pretrainer.py:667-680); the positive set
        is kept as the CSR graph instead."""
        return graph_of(adj)

    def sample_train(self, gt):
        smp = self._static_sampler(gt)
        idx, lab = smp.sample_padded() if sampling.PADDED_LISTS else smp.sample()
        return lab, [idx]

    def inference(self, data, sparse_edge_index=None):
        feature, adj = data
        return self.models[0].predict_adjs_sparse(feature, adj, [self.fuse1, self.fuse2],
                                                  auxiliary_edges=sparse_edge_index)

    def loss(self, data, labels, indices):
        pred_adjs = self.inference(data, indices)
        loss = None
        nh = self.args.nhead
        for i, pred_adj in enumerate(pred_adjs):
            if self._layer_on(i):
                term = pair_mse_loss([head[0] for head in pred_adj], 0, nh, labels, graph_of(data[1]))
                loss = term if loss is None else loss + term
        return loss

    def train_step(self, data, gt_adj=None):
        self._begin_step()
        labels, indices = self.sample_train(gt_adj if gt_adj is not None else data[1])
        loss = self.loss(data, labels, indices)
        self._finish_step(loss, graph_of(data[1]))
        return {"loss_heads_sup": loss.detach()}

    def _static_sampler(self, gt):
        return _graph_sampler(graph_of(gt))         # the positive set of a graph never changes: one sampler per graph

    def samplers(self, data, gt_adj=None):
        return [self._static_sampler(gt_adj if gt_adj is not None else data[1])]

    def _static_device(self, adam, data, gt_adj=None):
        indices, labels = self._static_sampler(gt_adj if gt_adj is not None else data[1]).sample_static()
        loss = self.loss(data, labels, [indices])
        self._static_finish(adam, loss)
        return {"loss_heads_sup": loss.detach()}


class GeneratedEdgeTrainer(Trainer):
    """Homo / hetero edge disentanglement: first H/2 heads vs same-label edges, last H/2 heads vs
    different-label edges (pretrainer.py:372-654)."""

    def __init__(self, args, model, weight):
        super().__init__(args, model, weight)
        self.dis_type = args.dis_type
        self.constrain_layer = args.constrain_layer
        self.sparse = args.sparse
        self.dis_adjs = []
        self.labels = None

    def get_label_all(self, feature, adj, labels, load=True):
        """Flat positive sets of the two edge groups (pretrainer.py:448-456) - O(E), nothing cached
        on disk (the reference pickles a dense 2 x N x N tensor).  `labels` are the GLOBAL node labels; on a row
        shard rows are offset by the shard's first row and columns are global already.  With --conformT only edges
        between nodes of the train + val split count as known (pretrainer.py:465-498; the split is drawn here from
        `random` exactly as the reference draws it)."""
        assert self.dis_type == 1, "currently only use homo&hetero edge disentanglement"
        g = graph_of(adj)
        self.labels = labels
        r_lab, c_lab = labels[g.row + g.row_start], labels[g.col.long()]
        same, diff = r_lab == c_lab, r_lab != c_lab
        if getattr(self.args, "conformT", False):
            idx_train, idx_val, _te, _m = split(labels.cpu(), train_ratio=self.args.node_sup_ratio)
            known = torch.zeros(labels.shape[0], dtype=torch.bool, device=labels.device)
            known[torch.cat((idx_train, idx_val)).to(labels.device)] = True
            both = known[g.row + g.row_start] & known[g.col.long()]
            same, diff = same & both, diff & both
        flat = sampling.flat_edges(g)
        self.graph = g
        self.dis_adjs = [flat[same], flat[diff]]
        self.n_pos_global = [_global_count(p, g) for p in self.dis_adjs]
        self._samplers = None
        return self.dis_adjs

    def sample_train(self):
        labs, idxs = [], []
        for smp in self._static_samplers():
            idx, lab = smp.sample_padded() if sampling.PADDED_LISTS else smp.sample()
            labs.append(lab)
            idxs.append(idx)
        return labs, idxs

    def inference(self, data, sparse_edge_index=None, head_ranges=None):
        feature, adj = data
        return self.models[0].predict_adjs_sparse(feature, adj, [self.fuse1, self.fuse2],
                                                  auxiliary_edges=sparse_edge_index, head_ranges=head_ranges)

    def loss(self, data, adj_labels, adj_masks):
        nh = self.args.nhead
        half = int(nh / 2)
        # only the supervised half of the heads is scored on each list (the reference scores all)
        pred_adjs = self.inference(data, adj_masks, head_ranges=[(0, half), (half, nh)])
        loss = None
        for i, pred_adj in enumerate(pred_adjs):
            if self._layer_on(i):
                g = graph_of(data[1])
                term = (pair_mse_loss([head[0] for head in pred_adj], 0, half, adj_labels[0], g)
                        + pair_mse_loss([head[1] for head in pred_adj], half, nh, adj_labels[1], g))
                loss = term if loss is None else loss + term
        return loss

    def train_step(self, data, pre_adjs=None):
        assert self.dis_type == 1, "currently only use homo&hetero edge disentanglement"
        self._begin_step()
        adj_labels, adj_masks = self.sample_train()
        loss = self.loss(data, adj_labels, adj_masks)
        self._finish_step(loss, graph_of(data[1]))
        return {"loss_head_disen": loss.detach()}

    def _static_samplers(self):
        if self.__dict__.get("_samplers") is None:
            self._samplers = [_sampler(self.graph, pos, n) for pos, n in zip(self.dis_adjs, self.n_pos_global)]
        return self._samplers

    def samplers(self, data=None, pre_adjs=None):
        return self._static_samplers()

    def _static_device(self, adam, data, pre_adjs=None):
        pairs = [smp.sample_static() for smp in self._static_samplers()]
        loss = self.loss(data, [lab for _idx, lab in pairs], [idx for idx, _lab in pairs])
        self._static_finish(adam, loss)
        return {"loss_head_disen": loss.detach()}


class DifHeadTrainer(Trainer):
    """Head-diversity: an MLP must recognise which head produced cat(layer_input, head_output)
    (pretrainer.py:780-857)."""

    def __init__(self, args, model, weight):
        super().__init__(args, model, weight)
        assert args.model == "DISGAT", "divergence on heads is only supported for DISGAT"
        dev = next(model.parameters()).device
        self.classifier1 = MLP(in_feat=args.nhid + args.size, hidden_size=args.nhid, out_size=args.nhead,
                               layers=args.cls_layer).to(dev)
        self.classifier2 = MLP(in_feat=args.nhid * 2, hidden_size=args.nhid, out_size=args.nhead,
                               layers=args.cls_layer).to(dev)
        for c in (self.classifier1, self.classifier2):
            self.models.append(c)
            self.models_opt.append(make_adam(c.parameters(), args))
        self.nhead = args.nhead

    def get_label_all(self, feature, adj):
        return None

    def inference(self, data):
        feature, adj = data
        return self.models[0].get_edge_em(feature, adj, [self.fuse1, self.fuse2])

    def loss(self, data):
        """sum over layers and heads of NLL(log_softmax(MLP(cat(layer_in, head_out_i))), i)
        (pretrainer.py:819-832).  cat(in, out_i) @ W1^T splits into in @ W1[:, :F_in]^T, shared by
        all heads of a layer, plus out_i @ W1[:, F_in:]^T, so the [N, F_in+nhid] concatenations the
        reference builds per head (models.py:347, 365) are never materialised here; get_edge_em()
        still returns them for callers that want them."""
        feature, adj = data
        r = self.models[0]._run(feature, adj, [self.fuse1, self.fuse2], heads_f32=False)
        g = graph_of(adj)
        sharded = isinstance(g, parallel.DistGraph) and g.world > 1
        loss = None
        for layer, (inp, heads) in enumerate(((r["x"], r["heads"][0]), (r["feature_1"], r["heads"][1]))):
            classifier = self.classifier1 if layer == 0 else self.classifier2
            mods = list(classifier.model)
            fused = getattr(heads, "fused", None)
            planes = getattr(heads, "planes", None)
            nh, fo = getattr(heads, "n_heads", 0) or len(heads), getattr(heads, "f_out", 0) or heads[0].shape[1]
            batched = (len(mods) >= 3 and isinstance(mods[0], torch.nn.Linear) and isinstance(mods[1], torch.nn.LeakyReLU)
                       and (planes is not None or (fused is not None and fused.shape[1] == nh * fo)))
            if fused is None and planes is not None and not (batched and not torch.is_grad_enabled()
                                                             and ops_gemm.planes_ok(fo, mods[0].out_features)):
                fused = planes.to_f32()            # a classifier shape the plane GEMM does not tile: fp32 heads after all
                heads = [fused[:, h * fo:(h + 1) * fo] for h in range(nh)]
            if len(mods) < 3 or not isinstance(mods[0], torch.nn.Linear):     # cls_layer == 1: no shared part
                outs = [classifier(torch.cat((inp, h), dim=-1), cls=True) for h in heads]
            elif batched:
                # all heads in one batched GEMM on the fused [N, H*nhid] buffer: no per-head slices in the autograd
                # graph (each slice's backward zero-fills and re-adds a full [N, H*nhid] gradient)
                lin = mods[0]
                f_in = inp.shape[1]
                shared = ops_gemm.linear(inp, lin.weight[:, :f_in].t(), lin.bias)
                w_h = lin.weight[:, f_in:].t().unsqueeze(0).expand(nh, fo, -1)
                if (fused is None and len(mods) == 3 and isinstance(mods[2], torch.nn.Linear)
                        and ops_gemm.logits_ok(fo, lin.out_features, mods[2].out_features)):
                    # no-graph forward, hidden -> logits in the hidden layer's epilogue (disgat_gemm_planes_logits): the
                    # [N * nhead, hidden] activations (8 GB at 1M nodes) are neither written nor read back
                    from .layers import _memo
                    w_rm, w2p, wnorm = _memo([classifier], "difhead_logits", lambda: (
                        ops_gemm.presplit_rm(w_h), ops_gemm.presplit_logits(mods[2].weight, mods[2].bias),
                        ops_gemm.weight_bound(lin.weight[:, f_in:].t())[0]))
                    mid_bound = (planes.bound.reshape(-1)[0] * wnorm * 1.001 + ops_gemm.amax(shared).reshape(-1)[0]).reshape(1)
                    t = ops_gemm.linear_planes_logits(planes.view_heads(nh), w_rm, None, shared, ops_gemm.ACT_LEAKY,
                                                      mods[1].negative_slope, mid_bound, w2p)
                    mods = mods[:2]         # nothing left to apply below
                elif fused is None:    # no-graph forward: the head buffer exists only as the GEMM's operand planes
                    t = ops_gemm.linear_planes(planes.view_heads(nh), ops_gemm.presplit_rm(w_h), lin.out_features, None, shared,
                                               ops_gemm.ACT_LEAKY, mods[1].negative_slope)[0]
                else:
                    t = ops_gemm.linear(fused.view(-1, nh, fo).permute(1, 0, 2), w_h, None, shared, ops_gemm.ACT_LEAKY,
                                        mods[1].negative_slope, a_amax=getattr(heads, "fused_amax", None))   # [N, H*hidden]
                if len(mods) > 2:
                    t = t.view(t.shape[0] * nh, -1)
                for m in mods[2:]:
                    # hidden -> nhead logits on N * nhead rows: 1 KB read per 32 B written (ops_gemm.skinny_linear)
                    t = ops_gemm.skinny_linear(t, m) if isinstance(m, torch.nn.Linear) and ops_gemm.skinny_ok(t, m) else m(t)
                # sum over heads i of NLL(log_softmax(t[(n, i)]), i), mean over nodes: row (n, i) of t carries label i - one
                # pass over the logits (ops.cls_loss) instead of log_softmax / diagonal / mean / neg and their backwards
                if t.stride(1) != 1:
                    t = t.contiguous()
                loc = ops.cls_loss(t, None, nh, g.n_global if sharded else t.shape[0] // nh)[0]
                term = loc + (parallel.all_reduce_sum(loc.detach().clone(), g) - loc.detach()) if sharded else loc
                loss = term if loss is None else loss + term
                continue
            else:
                lin = mods[0]
                f_in = inp.shape[1]
                shared = ops_gemm.linear(inp, lin.weight[:, :f_in].t(), lin.bias)
                w_h = lin.weight[:, f_in:].t()
                fuse_act = isinstance(mods[1], torch.nn.LeakyReLU)
                outs = []
                for h in heads:
                    if fuse_act:    # shared + h @ W + LeakyReLU in the GEMM epilogue
                        t = ops_gemm.linear(h, w_h, None, shared, ops_gemm.ACT_LEAKY, mods[1].negative_slope)
                        rest = mods[2:]
                    else:
                        t = ops_gemm.linear(h, w_h, None, shared)
                        rest = mods[1:]
                    for m in rest:
                        t = m(t)
                    outs.append(F.log_softmax(t, dim=1))
            for i, pred_label in enumerate(outs):
                if sharded:     # global mean over all ranks' nodes (value global, gradient = local part)
                    loc = -pred_label[:, i].sum() / g.n_global
                    term = loc + (parallel.all_reduce_sum(loc.detach().clone(), g) - loc.detach())
                else:
                    term = -pred_label[:, i].mean()                 # NLLLoss against the constant label i
                loss = term if loss is None else loss + term
        return loss

    def train_step(self, data, pre_dif=None):
        self._begin_step()
        loss = self.loss(data)
        self._finish_step(loss, graph_of(data[1]))
        return {"loss_head_diversity": loss.detach()}

    def _static_device(self, adam, data, pre_dif=None):
        loss = self.loss(data)
        self._static_finish(adam, loss)
        return {"loss_head_diversity": loss.detach()}
