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


def _worker(rank, world, port, q, golden_dir):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        from edgedisentangle_ssl_amd import parallel, pretrainer
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
        for gnn, att in (("AT", 3), ("SAGE", 1), ("GCN", 2)):
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
                for a_, b_ in zip(g_sh, g_ref):
                    if b_ is None:
                        assert a_ is None or float(a_.abs().max()) == 0.0
                        continue
                    err = (a_ - b_).abs().max().item()
                    assert err <= 2e-4 * max(1e-3, b_.abs().max().item()), (gnn, att, nm, err, b_.abs().max().item())
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_sharded_equals_unsharded_two_ranks_one_gpu(golden_dir):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, golden_dir)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"
