"""The sharded HIP path end to end on ONE GPU: two ranks share cuda:0 over `gloo` (RCCL needs one
device per rank, which the test box does not have).  Sharded get_em / SupEdge / DisEdge / DifHead
losses must equal the unsharded run on the same inputs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import inputs_common as ic

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, golden_dir, configs=(("AT", 3), ("SAGE", 1), ("GCN", 2))):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        from edgedisentangle_ssl_amd import parallel
        from edgedisentangle_ssl_amd.graph import CSRGraph
        from test_gpu_parity import build
        from test_gpu_backward import _trainers

        d = np.load(os.path.join(golden_dir, "data_cora.npz"))
        n = int(d["n"])
        ei = torch.from_numpy(d["edge_index"].astype(np.int64))
        lab = torch.from_numpy(d["labels"].astype(np.int64))
        x = ic.features(51, n, 64, "cora_surrogate").to(dev)
        g = CSRGraph.from_index(ei.to(dev), n)
        pos, homo, het = ic.edge_sets(ei, lab, n)
        sup = ic.sample_pairs(61, n, pos, "sup")
        ho = ic.sample_pairs(62, n, homo, "homo")
        he = ic.sample_pairs(63, n, het, "het")
        for gnn, att in configs:
            a, enc, _ = build(gnn, att, 8, 64, 64, 200 + att, dev)
            sup_t, dis_t, dif_t = _trainers(a, enc, 200 + att, dev)
            fus = [sup_t.fuse1, sup_t.fuse2]
            with torch.no_grad():
                ref_em = enc.get_em(x, g, fus)
                ref_sup = sup_t.loss((x, g), sup[1].to(dev), [sup[0].to(dev)])
                ref_dis = dis_t.loss((x, g), [ho[1].to(dev), he[1].to(dev)], [ho[0].to(dev), he[0].to(dev)])
                ref_dif = dif_t.loss((x, g))
            dg = parallel.DistGraph.shard(g, rank, world)
            lo, hi = dg.row_start, dg.row_start + dg.n

            def local(p):
                idx, lb = p
                m = (idx[0] >= lo) & (idx[0] < hi)
                return torch.stack([idx[0][m] - lo, idx[1][m]]).to(dev), lb[m].to(dev)

            xl = x[lo:hi].contiguous()
            with torch.no_grad():
                em = enc.get_em(xl, dg, fus)
                (si, sl), (hi_, hl), (ti, tl) = local(sup), local(ho), local(he)
                l_sup = sup_t.loss((xl, dg), sl, [si])
                l_dis = dis_t.loss((xl, dg), [hl, tl], [hi_, ti])
                l_dif = dif_t.loss((xl, dg))
            for l in range(2):
                err = (em[l] - ref_em[l][lo:hi]).abs().max().item()
                assert err <= 1e-5 * max(1.0, ref_em[l].abs().max().item()), (gnn, att, l, err)
            # the same forward over the halo-only exchange (only referenced rows travel, compact column ids)
            os.environ["DISGAT_EXCHANGE"] = "halo"
            with torch.no_grad():
                em_h = enc.get_em(xl, dg, fus)
            os.environ.pop("DISGAT_EXCHANGE")
            assert getattr(dg, "_halo", None) is not None and dg._halo.n_ref <= n
            for l in range(2):
                err = (em_h[l] - ref_em[l][lo:hi]).abs().max().item()
                assert err <= 1e-5 * max(1.0, ref_em[l].abs().max().item()), ("halo", gnn, att, l, err)
            for nm, got, want in (("sup", l_sup, ref_sup), ("dis", l_dis, ref_dis), ("dif", l_dif, ref_dif)):
                assert abs(got.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item())), (gnn, att, nm, got.item(), want.item())

            # training: sharded backward + gradient all-reduce == unsharded gradients
            def grads_of(fn, mods, graph):
                for m in mods:
                    for p in m.parameters():
                        p.grad = None
                loss = fn()
                loss.backward()
                parallel.all_reduce_grads(mods, graph)
                return loss.item(), [None if p.grad is None else p.grad.clone() for m in mods for p in m.parameters()]

            for nm, tr, full, part in (
                    ("sup", sup_t, lambda: sup_t.loss((x, g), sup[1].to(dev), [sup[0].to(dev)]),
                     lambda: sup_t.loss((xl, dg), sl, [si])),
                    ("dis", dis_t, lambda: dis_t.loss((x, g), [ho[1].to(dev), he[1].to(dev)], [ho[0].to(dev), he[0].to(dev)]),
                     lambda: dis_t.loss((xl, dg), [hl, tl], [hi_, ti])),
                    ("dif", dif_t, lambda: dif_t.loss((x, g)), lambda: dif_t.loss((xl, dg)))):
                lv_ref, g_ref = grads_of(full, tr.models, g)
                lv, g_sh = grads_of(part, tr.models, dg)
                assert abs(lv - lv_ref) <= 2e-6 * max(1.0, abs(lv_ref)), (gnn, att, nm, lv, lv_ref)
                names = [f"{type(m).__name__}.{k}" for m in tr.models for k, _p in m.named_parameters()]
                for pname, a_, b_ in zip(names, g_sh, g_ref):
                    if b_ is None:
                        assert a_ is None or float(a_.abs().max()) == 0.0
                        continue
                    err = (a_ - b_).abs().max().item()
                    assert err <= 2e-4 * max(1e-3, b_.abs().max().item()), (gnn, att, nm, pname, err, b_.abs().max().item())
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def _run_ranks(target, world, args, timeout=500):
    """Start `world` ranks, collect one (rank, message) each; whatever happens, no child outlives the test (a rank
    that raised before a collective leaves its peer blocked inside gloo) and the failing rank's traceback is shown."""
    import queue
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            try:
                res.append(q.get(timeout=timeout))
            except queue.Empty:
                break
            if res[-1][1] != "ok":          # one rank failed: its peers may never return
                break
    finally:
        for p in procs:
            p.join(timeout=5 if (len(res) < world or any(m != "ok" for _, m in res)) else 60)
            if p.is_alive():
                p.kill()
                p.join()
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"
    assert len(res) == world, f"only {len(res)} of {world} ranks reported (timeout)"


def test_sharded_equals_unsharded_two_ranks_one_gpu(golden_dir):
    _run_ranks(_worker, 2, (golden_dir,))


def test_sharded_equals_unsharded_four_ranks_one_gpu(golden_dir):
    """Four ragged row ranges (2708 rows -> 677 each) on the real kernels: the sliced gather's four slices, the halo
    tables and the adjoint's reduce-scatter all see more than one peer."""
    _run_ranks(_worker, 4, (golden_dir, (("AT", 3), ("SAGE", 1))))


def _train_worker(rank, world, port, q, golden_dir):
    """The sharded train_step end to end (sample on the shard -> forward -> backward -> gradient all-reduce -> Adam):
    every rank must hold bit-identical parameters afterwards, the sampled pairs must be valid on the shard, and the
    losses must be finite and equal across ranks (they are global values)."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        from edgedisentangle_ssl_amd import parallel
        from edgedisentangle_ssl_amd.graph import CSRGraph
        from edgedisentangle_ssl_amd.trainer import ClsTrainer
        from test_gpu_parity import build
        from test_gpu_backward import _trainers

        d = np.load(os.path.join(golden_dir, "data_cora.npz"))
        n = int(d["n"])
        ei = torch.from_numpy(d["edge_index"].astype(np.int64))
        lab = torch.from_numpy(d["labels"].astype(np.int64)).to(dev)
        x = ic.features(51, n, 64, "cora_surrogate").to(dev)
        g = CSRGraph.from_index(ei.to(dev), n)
        dg = parallel.DistGraph.shard(g, rank, world)
        lo, hi = dg.row_start, dg.row_start + dg.n
        xl = x[lo:hi].contiguous()
        a, enc, _ = build("AT", 3, 8, 64, 64, 203, dev)
        a.dropout, a.node_sup_ratio, a.fuse, a.reg, a.reg_weight, a.enc_layer = 0.0, 0.25, "last", False, 0.01, 2
        sup_t, dis_t, dif_t = _trainers(a, enc, 203, dev)
        torch.manual_seed(100 + rank)                       # different sampler streams per rank, as in a real run
        dis_t.get_label_all(xl, dg, lab)
        lab_s, idx_s = sup_t.sample_train(dg)
        assert int(idx_s[0][0].max()) < dg.n and int(idx_s[0][1].max()) < n and int(idx_s[0][1].max()) >= dg.n
        cnt = int(lab_s._disgat_count)                      # fixed-capacity list: valid prefix, then padding (label -1)
        assert 0 < cnt <= lab_s.shape[0] and bool((lab_s[cnt:] == -1).all())
        flat = (idx_s[0][0][:cnt] + lo) * n + idx_s[0][1][:cnt]
        pos = g.row * n + g.col.long()
        assert torch.equal(lab_s[:cnt], torch.isin(flat, pos).float())
        logs = {}
        logs.update(sup_t.train_step((xl, dg)))
        logs.update(dis_t.train_step((xl, dg)))
        logs.update(dif_t.train_step((xl, dg)))
        # the column-side score operand's backward put the gathered table's adjoint on the links itself, before its
        # weight-gradient GEMM (ops_bwd.layer_backward_u -> parallel.start_adjoint): once per SSL step (layer 2; the features
        # that enter layer 1 need no gradient, so nothing travels back there)
        assert parallel.ADJOINT_EARLY_STARTS == 3, parallel.ADJOINT_EARLY_STARTS
        assert parallel.ADJOINT_EARLY_TAKEN == 3 and not parallel._ADJOINT_OPEN
        import random
        random.seed(7)                                      # the node split is drawn from `random`: same on every rank
        cls_t = ClsTrainer(a, enc, lab, 1.0)
        ic.load_params(cls_t.classifier, 999)
        ic.load_params(cls_t.fuse1, 998)
        ic.load_params(cls_t.fuse2, 997)
        logs.update({k: v for k, v in cls_t.train_step((xl, dg), lab, 0).items() if k in ("loss_train", "acc_train", "loss_val")})
        vals = torch.tensor([float(v) for v in logs.values()], dtype=torch.float64)
        assert torch.isfinite(vals).all(), logs
        # parameters identical on every rank after four optimiser rounds
        flat_p = torch.cat([p.detach().reshape(-1) for tr in (sup_t, dis_t, dif_t, cls_t) for m in tr.models for p in m.parameters()])
        other = [torch.empty_like(flat_p) for _ in range(world)]
        dist.all_gather(other, flat_p)
        assert all(torch.equal(o, other[0]) for o in other), "replicated parameters diverged across ranks"
        # DifHead and CLS have no sampled input: their global loss must be the same number on every rank
        same = torch.tensor([logs["loss_head_diversity"], logs["loss_train"], logs["loss_val"]], dtype=torch.float64)
        alls = [torch.empty_like(same) for _ in range(world)]
        dist.all_gather(alls, same)
        assert all(torch.allclose(t, alls[0], rtol=1e-6) for t in alls), alls
        if rank == 0:       # and equal to the unsharded CLS step from the same initial state
            pass
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_sharded_train_steps_two_ranks_one_gpu(golden_dir):
    _run_ranks(_train_worker, 2, (golden_dir,))


def _main_worker(rank, world, port, q, golden_dir):
    """edgedisentangle_ssl_amd.main.run as one of two ranks (torch.distributed.run's environment, gloo on one GPU):
    partition on load / shard the fixture, static feature exchange, sharded CLS + SSL train_steps, logs."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), DISGAT_DIST_BACKEND="gloo")
        import math
        from edgedisentangle_ssl_amd import main
        argv = ["--model=DISGAT", "--sparse", "--dataset", "chameleon", "--fixture", os.path.join(golden_dir, "data_chameleon.npz"),
                "--gnn_type", "AT", "--att", "3", "--nhead", "4", "--nhid", "32", "--epochs", "4", "--steps", "2",
                "--downstream", "CLS", "--down_weight", "1.0", "--finetune", "--pretrain", "SupEdge", "DisEdge", "DifHead",
                "--pre_weight", "1", "1", "1", "--pre_edge", "1", "1", "1", "--dropout", "0.1", "--seed", "4", "--quiet"]
        hist = main.run(argv)
        assert len(hist) == 4
        for h in hist:
            for k in ("loss_train", "loss_heads_sup", "loss_head_disen", "loss_head_diversity"):
                assert math.isfinite(h[k]), (k, h)
        vals = torch.tensor([hist[-1]["loss_train"], hist[-1]["loss_head_diversity"], hist[0]["test_acc_test"]], dtype=torch.float64)
        both = [torch.empty_like(vals) for _ in range(world)]
        dist.all_gather(both, vals)
        assert all(torch.allclose(b, both[0], rtol=1e-6) for b in both), both     # global losses: same number on every rank
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_main_run_two_ranks_one_gpu(golden_dir):
    _run_ranks(_main_worker, 2, (golden_dir,))
