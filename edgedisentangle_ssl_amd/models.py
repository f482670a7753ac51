"""DISGAT encoder and MLP head: drop-in mirrors of /root/reference/models.py:151-373 and :523-543.

Same constructor and method signatures, same registered sub-module names
(`attention1_i`, `attention2_i`, `fuser1`, `fuser2` -> identical state_dict keys,
models.py:165-179), same return conventions.  The H per-head DisGALayer calls of
each layer run as one fused HIP edge pass (layers.disga_heads).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .layers import DisGALayer, FuseLayer, disga_heads


class DISGAT(nn.Module):
    def __init__(self, args, nfeat, nhid, nclass, dropout, is_specific=[True, True], alpha=0.1, nheads=4):
        super().__init__()
        self.dropout = dropout
        self.args = args
        self.nheads = nheads
        self.gnn_type = args.gnn_type
        self.is_specific = is_specific
        # predict_adjs_sparse returns only the aux scores; the reference nevertheless aggregates layer 2
        # and runs its fuser, then discards both (models.py:319-330).  Set True to skip that unobservable
        # work (outputs, gradients and losses are unchanged; off by default so timings compare like with like).
        self.skip_unused = False
        # parameter creation order (= torch RNG draw order of the reference, models.py:163-179): all heads of
        # layer 1, all heads of layer 2, then the model's own two fusers
        self.attentions1 = self._make_heads("attention1_", nfeat, nhid, dropout, alpha, args.att)
        self.attentions2 = self._make_heads("attention2_", nhid, nclass, dropout, alpha, args.att)
        res = (nfeat, nhid) if args.residue else (0, 0)
        self.fuser1 = FuseLayer(args, nheads, nfeat=nhid, residue=res[0])
        self.fuser2 = FuseLayer(args, nheads, nfeat=nhid, residue=res[1])

    def _make_heads(self, prefix, fin, fout, dropout, alpha, att):
        """H DisGALayer heads registered as `<prefix><i>` (the reference's state_dict keys)."""
        heads = []
        for i in range(self.nheads):
            head = DisGALayer(fin, fout, dropout=dropout, alpha=alpha, concat=True, att_type=att, gnn_type=self.gnn_type)
            self.add_module(prefix + str(i), head)
            heads.append(head)
        return heads

    # The five entry points of the reference share one two-layer loop (models.py:181-373);
    # `_run` is that loop, returning everything any of them needs.
    def _run(self, x, adj, fusers, auxiliary_edges=None, head_ranges=None, scores_only_layer2=False, heads_f32=True,
             discard_layer2=False, heads_to_fuser_only=False):
        """heads_f32=False: the caller reads the per-head outputs only through HeadList.planes / .fused (get_em, the
        score entry points, DifHead's batched classifier) - a layer whose fuser takes planes then never writes the
        fp32 [N, H*nhid] head buffer on a no-graph forward.  heads_to_fuser_only: nothing but the layer's fuser reads the
        heads (every entry point except get_edge_em / DifHead): the projection is then left to the fuser, which may run it
        back to back with its own GEMM (layers.DeferredHeads) - no head buffer in any form."""
        if not isinstance(fusers, list):
            fusers = [fusers]
        fu1 = self.fuser1 if not self.is_specific[0] else fusers[0]
        fu2 = self.fuser2 if not self.is_specific[1] else fusers[1]

        def planes_only(fuser):
            return (not heads_f32 and not torch.is_grad_enabled() and isinstance(fuser, FuseLayer) and fuser.accepts_planes())

        x = F.dropout(x, self.dropout, training=self.training)
        h1, adj1, aux1 = disga_heads(self.attentions1, x, adj, auxiliary_edges, head_ranges, heads_planes=planes_only(fu1),
                                     heads_deferrable=heads_to_fuser_only, defer_join=True)
        f1 = fu1(h1, x)
        feature_1 = F.dropout(f1, self.dropout, training=self.training)
        # discard_layer2: the caller drops layer 2's heads / scores / fuser output (predict_adjs_sparse; the reference
        # computes them, models.py:319-330): computed as values only - no sign record, no graph behind them
        h2, adj2, aux2 = disga_heads(self.attentions2, feature_1, adj, auxiliary_edges, head_ranges,
                                     aux_only=scores_only_layer2, heads_planes=planes_only(fu2),
                                     heads_discarded=discard_layer2, heads_deferrable=heads_to_fuser_only, defer_join=True)
        if scores_only_layer2:
            ops.join_side()
            return dict(x=x, feature_1=feature_1, x2=None, heads=(h1, None), adjs=(adj1, None), aux=(aux1, aux2))
        if discard_layer2:
            with torch.no_grad():
                f2 = fu2(h2, feature_1)
        else:
            f2 = fu2(h2, feature_1)
        ops.join_side()                 # the pair scores of both layers (run beside the layers' GEMMs on the side stream)
        return dict(x=x, feature_1=feature_1, x2=f2, heads=(h1, h2), adjs=(adj1, adj2), aux=(aux1, aux2))

    def forward(self, x, adj, fusers):                                   # models.py:181-214
        return F.log_softmax(self._run(x, adj, fusers, heads_f32=False, heads_to_fuser_only=True)["x2"], dim=1)

    def get_em(self, x, adj, fusers):                                    # models.py:217-252
        r = self._run(x, adj, fusers, heads_f32=False, heads_to_fuser_only=True)
        feature_2 = F.dropout(r["x2"], self.dropout, training=self.training)
        return [r["feature_1"], feature_2]

    def get_adjs(self, x, adj, fusers):                                  # models.py:254-288
        r = self._run(x, adj, fusers, heads_f32=False, heads_to_fuser_only=True)
        return [r["adjs"][0], r["adjs"][1]]

    def predict_adjs_sparse(self, x, adj, fusers, auxiliary_edges, head_ranges=None):   # models.py:290-330
        r = self._run(x, adj, fusers, auxiliary_edges, head_ranges, scores_only_layer2=self.skip_unused, heads_f32=False,
                      discard_layer2=True, heads_to_fuser_only=True)
        return [r["aux"][0], r["aux"][1]]

    def get_edge_em(self, x, adj, fusers):                               # models.py:333-373
        r = self._run(x, adj, fusers)
        e1 = [torch.cat((r["x"], h), dim=-1) for h in r["heads"][0]]
        e2 = [torch.cat((r["feature_1"], h), dim=-1) for h in r["heads"][1]]
        return [e1, e2]


class MLP(nn.Module):
    """models.py:523-543: Linear -> LeakyReLU(0.1) -> ... -> Linear (+ log_softmax when cls)."""

    def __init__(self, in_feat, hidden_size, out_size, layers=2, dropout=0.1):
        super().__init__()
        modules = []
        in_size = in_feat
        for _ in range(layers - 1):
            modules.append(nn.Linear(in_size, hidden_size))
            in_size = hidden_size
            modules.append(nn.LeakyReLU(0.1))
        modules.append(nn.Linear(in_size, out_size))
        self.model = nn.Sequential(*modules)

    def forward(self, features, cls=False):
        output = self.model(features)
        return F.log_softmax(output, dim=1) if cls else output
