"""`python -m edgedisentangle_ssl_amd.main ...`: the reference's main.py flow for --model=DISGAT
(main.py:24-365): same flags, same alternation of downstream and SSL train_steps per epoch.

Extra flags (not in the reference): --data_root (directory holding <dataset>/ in the reference's
format) or --fixture (a tests/golden/data_<name>.npz file), --quiet.
"""
import os
import random
import sys
import time

import numpy as np
import torch

from . import data_load, models, pretrainer, trainer
from .features import surrogate_features
from .graph import graph_of
from .utils import get_parser, resolve_logs

CAPTURE_MAX_EDGES = 2_000_000     # --capture auto: above this a step is kernel-bound and the eager path's load balancing matters
CAPTURE_MIN_EPOCHS = 24           # --capture auto: capturing costs 0.3-0.5 s once; at ~20 ms saved per epoch shorter runs stay eager

SSL = {"DisEdge": pretrainer.GeneratedEdgeTrainer, "SupEdge": pretrainer.SupEdgeTrainer, "DifHead": pretrainer.DifHeadTrainer}


def checkpoint_path(args, epoch, root="."):
    """The reference's checkpoint location (main.py:218, 230): one file per (pretrain list, epoch)."""
    folder = "{}/checkpoint/{}/{}_used_edge{}_weight{}_reg{}".format(root, args.dataset, args.model, args.used_edge,
                                                                   args.pre_weight, args.reg)
    return folder, "{}/pretrain_{}_{}.pth".format(folder, args.pretrain, epoch)


def save_model(encoder, args, epoch, root="."):
    """main.py:214-227: {'encoder': state_dict} - the keys are the reference's, so either side loads the other's file."""
    folder, path = checkpoint_path(args, epoch, root)
    os.makedirs(folder, exist_ok=True)
    torch.save({"encoder": encoder.state_dict()}, path)
    return path


def load_model(encoder, args, root="."):
    """main.py:229-235: restore the encoder saved at epoch `--load` under the same settings."""
    _, path = checkpoint_path(args, args.load, root)
    content = torch.load(path, map_location=lambda storage, loc: storage)
    encoder.load_state_dict(content["encoder"])
    return path


def reseed_rank(seed, rank):
    """Sharded runs: parameter initialisation and utils.split() must draw the SAME numbers on every rank (replicated
    parameters, global index sets) and have done so by now; from here on every rank needs its OWN streams - the pair
    sampler (sampling.PairSampler takes its seed from torch's CPU generator at first use), the input dropout and the in-kernel attention-dropout seed
    (taken from torch's CPU generator, layers.disga_heads) would otherwise repeat the same (local row, column)
    negatives and the same per-edge mask on every equal-size shard, and the union over ranks would not be the unsharded
    Bernoulli(3 rho) sample sampling.py promises."""
    s = int(seed) + 1000003 * (int(rank) + 1)
    random.seed(s)
    np.random.seed(s % (2 ** 32))
    torch.manual_seed(s)
    torch.cuda.manual_seed(s)


def run(argv=None, log=print):
    parser = get_parser()
    parser.add_argument("--data_root", type=str, default="data")
    parser.add_argument("--fixture", type=str, default=None)
    parser.add_argument("--quiet", action="store_true", default=False)
    parser.add_argument("--capture", choices=("auto", "on", "off"), default="auto",
                        help="replay every train_step from a HIP graph (capture.StaticStep); auto = one process, a graph of at "
                             "most CAPTURE_MAX_EDGES edges (where a step is launch-bound) and at least CAPTURE_MIN_EPOCHS epochs "
                             "(capturing costs 0.3-0.5 s once per run)")
    args = parser.parse_args(argv)
    if args.model != "DISGAT":
        raise SystemExit("only --model=DISGAT is implemented by this package (SURVEY 2: other encoders out of scope)")
    for flag, why in (("batch", "sub-graph mini-batching (dataset.py) is outside the DISGAT hot path (SURVEY 2 row 22)"),
                      ("hnn", "the heterogeneous-network encoders are outside the DISGAT hot path (SURVEY 2 row 17)")):
        if getattr(args, flag):
            raise SystemExit("--{}: {}".format(flag, why))
    if not torch.cuda.is_available() or args.no_cuda:
        raise SystemExit("the DISGAT HIP path needs an MI355X; there is no CPU fallback")
    args.cuda = True
    args.hetero = True                                  # main.py:29-30 (pretrain given) - adjs is a list
    args.sparse = True
    random.seed(args.seed)                              # main.py:44-48
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed)
    if args.pre_weight is None:
        args.pretrain = []
    unsupported = [p for p in (args.pretrain or []) if p not in SSL]
    if unsupported:
        raise SystemExit("SSL tasks {} are outside the DISGAT hot path (SURVEY 2 row 19)".format(unsupported))

    # one process per GPU (torch.distributed.run sets RANK / WORLD_SIZE / LOCAL_RANK): the graph is partitioned by
    # row ranges on load, every rank keeps its own rows of the features, labels stay global
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    backend = os.environ.get("DISGAT_DIST_BACKEND", "nccl")      # "gloo": several ranks sharing one GPU (tests / rehearsal)
    local = 0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local if world > 1 else torch.cuda.current_device())
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "gloo":
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
    if args.fixture:
        adj, features, labels = data_load.load_fixture(args.fixture)
        if features is None:
            features = surrogate_features(adj.shape[0], 64, seed=51)
        adjs = [adj.to(dev)]
        features = features.to(dev)
        if world > 1:
            from .graph import CSRGraph
            from .parallel import DistGraph
            adjs = [DistGraph.shard(CSRGraph.from_adj(adjs[0]), rank, world)]
            features = features[adjs[0].row_start: adjs[0].row_start + adjs[0].n].contiguous()
    else:
        args.edge_num = 1
        adjs, features, labels = data_load.load_data(args, path="{}/{}/".format(args.data_root, args.dataset),
                                                     dataset=args.dataset, edge_type=args.edge_num,
                                                     rank=rank, world=world, device=dev)
        adjs = [a.to(dev) if torch.is_tensor(a) else a for a in adjs]
        features = features.to(dev)
    args.size = features.shape[1]
    args.nclass = labels.max().item() + 1
    labels = labels.to(dev)
    if world > 1:
        from . import parallel
        parallel.mark_static(features)       # exchanged once, reused by every pass that does not drop out its input

    encoder = models.DISGAT(args, nfeat=args.size, nhid=args.nhid, nclass=args.nhid, nheads=args.nhead,
                            dropout=args.dropout).to(dev)                  # main.py:142-147
    # SupEdge / DisEdge only read the aux scores of predict_adjs_sparse: skip the layer-2 aggregation + fuser the
    # reference computes and discards there (same losses and gradients, ~12 % less work per SSL step)
    encoder.skip_unused = True
    if args.load is not None:                                             # main.py:262-263
        load_model(encoder, args)
    ssl_trainers, ssl_labels = [], []
    for i, name in enumerate(args.pretrain or []):                        # main.py:242-251
        assert args.pre_edge[i] > 0, "edge index begins from 1"
        tr = SSL[name](args, encoder, args.pre_weight[i])
        a = adjs[args.used_edge - 1]
        ssl_labels.append(tr.get_label_all(features, a, labels) if name == "DisEdge" else tr.get_label_all(features, a))
        ssl_trainers.append(tr)
    down = []
    for i, name in enumerate(args.downstream or []):                      # main.py:255-258
        if name != "CLS":
            raise SystemExit("downstream 'Edge' is declared unfinished by the reference (README.md:23)")
        down.append(trainer.ClsTrainer(args, encoder, labels, args.down_weight[0]))

    if world > 1:
        reseed_rank(args.seed, rank)
    captured = args.capture == "on" or (args.capture == "auto" and world == 1 and args.epochs >= CAPTURE_MIN_EPOCHS
                                        and all(graph_of(a).nnz <= CAPTURE_MAX_EDGES for a in adjs))
    if captured and world > 1:
        raise SystemExit("--capture on: captured steps run on one process (the sharded step holds collectives)")
    if captured:
        adjs = [graph_of(a) for a in adjs]               # the same objects every step: a captured step replays on them
    data_of = {}                                         # one [features, adj] list per adjacency, for the same reason
    history = []
    t0 = time.time()
    for epoch in range(args.epochs):                                      # main.py:270-360
        adj = adjs[args.used_edge - 1]
        log_ep = {"epoch": epoch}
        if epoch % 40 == 0:
            for i, tr in enumerate(down):
                log_ep.update({"test_" + k: v for k, v in tr.test([features, adj], labels, epoch).items()})
            if args.case and down:
                # main.py:289-301: head-to-head score correlation of the two layers (the reference also draws the grids
                # into tensorboard; Trainer.analyze_disentangle returns them)
                if world > 1:
                    raise SystemExit("--case: the disentanglement study runs on one process")
                dist, _at, _feat = down[-1].analyze_disentangle(features, adj)
                log_ep.update({"att_correlation_layer1": dist[0], "att_correlation_layer2": dist[1]})
        if args.finetune:
            for _ in range(args.steps):
                for tr in down:
                    if captured:
                        log_ep.update(tr.train_step_captured(data_of.setdefault(id(adj), [features, adj]), labels))
                    else:
                        log_ep.update(tr.train_step([features, adj], labels, epoch))
        for i, tr in enumerate(ssl_trainers):
            a = adjs[args.pre_edge[i] - 1]
            if captured:
                log_ep.update(tr.train_step_captured(data_of.setdefault(id(a), [features, a]), ssl_labels[i]))
            else:
                log_ep.update(tr.train_step([features, a], ssl_labels[i]))
        # the pair samplers' event counters (sampling.PairSampler: an item over a kernel capacity / a fixed-capacity list that
        # came out longer than its 8-sigma capacity and was cut) travel with the step logs; a run that trained on a cut list says so
        events = [smp.meta[4:6] for i, tr in enumerate(ssl_trainers) for smp in
                  tr.samplers([features, adjs[args.pre_edge[i] - 1]], ssl_labels[i]) if hasattr(smp, "meta")]
        if events:
            ev = torch.stack(events).sum(0)
            log_ep["sampler_overflow"], log_ep["sampler_clamped"] = ev[0], ev[1]
        log_ep = resolve_logs(log_ep)                    # the step logs are device scalars: one transfer per epoch
        if events:
            bad = (int(log_ep.pop("sampler_overflow")), int(log_ep.pop("sampler_clamped")))
            if bad != (0, 0):
                msg = "pair sampler: {} item(s) over a kernel capacity, {} list(s) cut at the fixed capacity".format(*bad)
                if captured:
                    raise RuntimeError(msg + " (--capture on trains on fixed-capacity lists; rerun with --capture off)")
                log_ep["sampler_events"] = msg
        history.append(log_ep)
        if not args.quiet:
            log(" ".join("{}={:.5g}".format(k, v) if isinstance(v, float) else "{}={}".format(k, v) for k, v in log_ep.items()))
    if not args.quiet:
        log("Optimization Finished!  Total time elapsed: {:.4f}s".format(time.time() - t0))
    return history


if __name__ == "__main__":
    run(sys.argv[1:])
