"""`gloo` tests (CPU) of the row-range sharding at world sizes 2, 4 and 8 (BASELINE's configs[3] / [4] run on 4 / 8 ranks):
partition covers every edge once, the per-layer all-gather (equal and ragged row counts - at 8 ranks some own a handful of
rows, slices are ragged and padded - with autograd), the adjoint started early by its producer, the halo exchange, the
gradient bucket with ranks that produced no gradient, and the global loss reduction.  The edge math itself is stood in for
by the oracle (tests only)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import inputs_common as ic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from edgedisentangle_ssl_amd import parallel
        from edgedisentangle_ssl_amd.graph import CSRGraph
        from oracle import disgat_oracle as orc
        from test_oracle_golden import shapes_layer

        idx, vals, n = ic.tiny_graph()
        g = CSRGraph.from_index(idx, n)
        dg = parallel.DistGraph.shard(g, rank, world)
        assert sum(dg.counts) == n and dg.n == dg.counts[rank]
        # every edge owned exactly once, in global CSR order
        nnz = torch.tensor([dg.nnz])
        dist.all_reduce(nnz)
        assert int(nnz) == g.nnz
        lo = dg.row_start
        e0 = int(g.rowptr[lo])
        assert torch.equal(dg.col, g.col[e0:e0 + dg.nnz])
        assert torch.equal(dg.row + lo, g.row[e0:e0 + dg.nnz])
        # balanced by nnz, not by rows
        assert abs(dg.nnz - g.nnz / world) <= g.nnz * 0.25 + 45

        # all-gather of the layer input: ragged counts + autograd (sum of grads lands on the owner)
        x = ic.features(21, n, 16)
        xl = x[lo:lo + dg.n].clone().requires_grad_(True)
        xa = parallel.all_gather_rows(xl, dg)
        assert torch.equal(xa.detach(), x)
        w = torch.arange(n, dtype=torch.float32).unsqueeze(1) * (rank + 1)
        (xa * w).sum().backward()
        want = (torch.arange(n, dtype=torch.float32).unsqueeze(1) * 3.0).expand(n, 16)[lo:lo + dg.n]
        assert torch.allclose(xl.grad, want)
        # equal counts -> all_gather_into_tensor path
        eq = parallel.DistGraph(4, dg.rowptr[:5], dg.col, dg.row, 4 * world, 4 * rank, [4] * world)
        xe = parallel.all_gather_rows(torch.full((4, 3), float(rank)), eq)
        assert torch.equal(xe, torch.arange(world, dtype=torch.float32).repeat_interleave(4).unsqueeze(1).expand(-1, 3))

        # pipelined exchange: the gather is handed over while its slices are still in flight; the column-side projection
        # consumes them as they land (own rows first), its adjoint reduce-scatters grad x back to the owners
        for slices in ("1", "3", "64"):
            os.environ["DISGAT_EXCHANGE_SLICES"] = slices
            xp = x[lo:lo + dg.n].clone().requires_grad_(True)
            wq = ic.features(5, 16, 24).requires_grad_(True)
            xg_, g_eff = parallel.exchange(xp, dg, edge_only=False, pipelined=True)
            assert g_eff is dg and parallel.pending_of(xg_) is not None
            qq = parallel.project_gathered(xg_, wq)
            assert parallel.pending_of(xg_) is None and torch.equal(xg_.detach(), x)
            assert torch.allclose(qq.detach(), x @ wq.detach(), atol=1e-5)
            cw = torch.cos(torch.arange(n * 24, dtype=torch.float32).view(n, 24)) * (rank + 1)
            ((qq * cw).sum() + (xg_ * w).sum()).backward()
            cw_tot = torch.cos(torch.arange(n * 24, dtype=torch.float32).view(n, 24)) * 3.0
            want_x = (cw_tot @ wq.detach().t() + torch.arange(n, dtype=torch.float32).unsqueeze(1) * 3.0)[lo:lo + dg.n]
            assert torch.allclose(xp.grad, want_x, atol=1e-4)
            assert torch.allclose(wq.grad, x.t() @ (cw_tot / 3.0 * (rank + 1)), atol=1e-3)
            # finish() without a projection (att 1 / 2: the table is only gathered from) - the late fills of the buffer
            # must not cut the other ranks' rows out of the gather's backward
            xq = x[lo:lo + dg.n].clone().requires_grad_(True)
            xg2, _ = parallel.exchange(xq, dg, edge_only=False, pipelined=True)
            assert torch.equal(parallel.finish(xg2).detach(), x) and parallel.pending_of(xg2) is None
            (xg2 * w).sum().backward()
            assert torch.allclose(xq.grad, want)
        os.environ.pop("DISGAT_EXCHANGE_SLICES")
        xe2, _ = parallel.exchange(torch.full((4, 3), float(rank)), eq, edge_only=False, pipelined=True)     # equal counts
        assert torch.equal(parallel.finish(xe2), xe)

        # one sharded layer (oracle as the math) == unsharded layer on the owned rows
        p = ic.make_params(shapes_layer("SAGE", 3, 16, 16), 9)
        ei_local = torch.stack([dg.row + lo, dg.col.long()])
        h_loc, e_loc, _ = orc.disga_layer(xa.detach(), ei_local, p, 3, "SAGE")
        h_all, e_all, _ = orc.disga_layer(x, g.indices(), p, 3, "SAGE")
        assert torch.allclose(h_loc[lo:lo + dg.n], h_all[lo:lo + dg.n], atol=1e-6)
        assert torch.allclose(e_loc, e_all[e0:e0 + dg.nnz], atol=1e-6)

        # halo-only exchange: just the referenced rows travel; the compact graph indexes the received table
        os.environ["DISGAT_EXCHANGE"] = "halo"
        xh = x[lo:lo + dg.n].clone().requires_grad_(True)
        x_ref, gc = parallel.exchange(xh, dg, edge_only=True)
        plan = dg._halo
        assert torch.equal(plan.ref, torch.unique(dg.col.long())) and x_ref.shape[0] == plan.n_ref == gc.n_cols
        assert torch.equal(x_ref.detach(), x[plan.ref]) and torch.equal(plan.ref[gc.col.long()], dg.col.long())
        assert gc.n == dg.n and torch.equal(gc.rowptr, dg.rowptr)
        # column-side score operand taken from the compact table == taken from the global one (att 3: Q = x W_bot)
        q_ref, q_all = x_ref.detach() @ p["W"][16:], x @ p["W"][16:]
        assert torch.equal(q_ref[gc.col.long()], q_all[dg.col.long()])
        wgt = torch.arange(n, dtype=torch.float32).unsqueeze(1) * (rank + 1)
        (x_ref * wgt[plan.ref]).sum().backward()                    # adjoint: owners sum what every rank sends back
        refd = torch.zeros(n)
        refd[plan.ref] = 1.0
        cnt = refd * (rank + 1)
        dist.all_reduce(cnt)                                        # sum over ranks of (rank+1) * [row referenced there]
        want_h = (torch.arange(n, dtype=torch.float32) * cnt).unsqueeze(1).expand(n, 16)[lo:lo + dg.n]
        assert torch.allclose(xh.grad, want_h)
        # an exchange for a pass with auxiliary pairs is always the full all-gather; "auto" decides by the referenced share
        xa2, g2 = parallel.exchange(xl.detach(), dg, edge_only=False)
        assert g2 is dg and torch.equal(xa2, x)
        os.environ["DISGAT_EXCHANGE"] = "allgather"
        xa3, g3 = parallel.exchange(xl.detach(), dg, edge_only=True)
        assert g3 is dg and torch.equal(xa3, x)
        os.environ.pop("DISGAT_EXCHANGE")
        # a tensor marked static is gathered once: the second exchange is served from the graph, until it changes
        xs = x[lo:lo + dg.n].clone()
        parallel.mark_static(xs)
        g1, _ = parallel.exchange(xs, dg, edge_only=False)
        g2, _ = parallel.exchange(xs, dg, edge_only=False)
        assert g2.data_ptr() == g1.data_ptr() and torch.equal(g1, x)            # same buffer: nothing was exchanged
        xs.mul_(2.0)                                                # version bump: the stale copy must not be served
        g3, _ = parallel.exchange(xs, dg, edge_only=False)
        assert g3.data_ptr() != g1.data_ptr() and torch.equal(g3, x * 2.0)
        xr = xs.clone().requires_grad_(True)                        # never for tensors a gradient is asked of
        parallel.mark_static(xr)
        g4, _ = parallel.exchange(xr, dg, edge_only=False)
        assert g4.requires_grad and dg._static_gather[1].data_ptr() != g4.data_ptr()

        # global pair loss from per-rank partial sums (what pretrainer.pair_mse_loss reduces)
        pairs, lab = ic.sample_pairs(31, n, np.sort((g.row * n + g.col.long()).numpy()), "sup")
        mine = (pairs[0] >= lo) & (pairs[0] < lo + dg.n)
        pred = torch.sigmoid(torch.sin(pairs[0].float() * 0.37 + pairs[1].float()))
        d2 = (pred - lab) ** 2
        acc = torch.tensor([float(d2[mine & (lab != 0)].sum()), float(d2[mine & (lab == 0)].sum()),
                            float((lab[mine] != 0).sum()), float(mine.sum())], dtype=torch.float64)
        parallel.all_reduce_sum(acc, dg)
        m = acc[3]
        loss = (acc[0] + acc[2] / (m * m - acc[2]) * acc[1]) / m
        assert abs(float(loss) - float(orc.adj_mse_loss(pred, lab))) < 1e-7
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def _worker_many(rank, world, port, q):
    """The exchange paths at any world size: ragged nnz-balanced ranges (ranks of very different row counts), 4 slices with
    per-slice padding, the adjoint, the adjoint put on the links early by the consumer's backward, the halo exchange, and
    all_reduce_grads when only some ranks produced a gradient."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from edgedisentangle_ssl_amd import parallel
        from edgedisentangle_ssl_amd.graph import CSRGraph
        idx, vals, n = ic.tiny_graph()
        g = CSRGraph.from_index(idx, n)
        dg = parallel.DistGraph.shard(g, rank, world)
        counts = dg.counts
        assert sum(counts) == n and dg.n == counts[rank] and len(counts) == world
        if world >= 4:
            assert len(set(counts)) > 1                     # ragged: the hub row's rank owns few rows
        nnz = torch.tensor([dg.nnz])
        dist.all_reduce(nnz)
        assert int(nnz) == g.nnz
        lo = dg.row_start
        tot = world * (world + 1) / 2.0                      # sum over ranks of (rank + 1)
        x = ic.features(21, n, 16)
        w = torch.arange(n, dtype=torch.float32).unsqueeze(1)
        want = (w * tot).expand(n, 16)[lo:lo + dg.n]
        for slices in ("1", "4", "64"):
            os.environ["DISGAT_EXCHANGE_SLICES"] = slices
            xp = x[lo:lo + dg.n].clone().requires_grad_(True)
            wq = ic.features(5, 16, 24).requires_grad_(True)
            xg_, g_eff = parallel.exchange(xp, dg, edge_only=False, pipelined=True)
            assert g_eff is dg
            if world > 1 and max(counts) > 0 and any(c for r, c in enumerate(counts) if r != rank):
                assert parallel.pending_of(xg_) is not None
            qq = parallel.project_gathered(xg_, wq)
            assert parallel.pending_of(xg_) is None and torch.equal(xg_.detach(), x)
            assert torch.allclose(qq.detach(), x @ wq.detach(), atol=1e-5)
            cw = torch.cos(torch.arange(n * 24, dtype=torch.float32).view(n, 24))
            ((qq * cw * (rank + 1)).sum() + (xg_ * w * (rank + 1)).sum()).backward()
            want_x = ((cw * tot) @ wq.detach().t() + w * tot)[lo:lo + dg.n]
            assert torch.allclose(xp.grad, want_x, atol=2e-4 * world)
            assert torch.allclose(wq.grad, x.t() @ (cw * (rank + 1)), atol=1e-3)
            xq = x[lo:lo + dg.n].clone().requires_grad_(True)
            xg2, _ = parallel.exchange(xq, dg, edge_only=False, pipelined=True)
            assert torch.equal(parallel.finish(xg2).detach(), x) and parallel.pending_of(xg2) is None
            (xg2 * w * (rank + 1)).sum().backward()
            assert torch.allclose(xq.grad, want)
        os.environ.pop("DISGAT_EXCHANGE_SLICES")

        # the adjoint started by its producer: a consumer whose backward finishes grad x_all, puts the reduce-scatter on the
        # links (parallel.start_adjoint), goes on with other work (here: its weight gradient) and returns; the gather's own
        # backward then only collects the result - same gradients as the plain path
        class Consumer(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x_all, wgt):
                ctx.save_for_backward(x_all, wgt)
                ctx.x_all = x_all
                return x_all @ wgt

            @staticmethod
            def backward(ctx, go):
                x_all, wgt = ctx.saved_tensors
                gx = go @ wgt.t()
                assert parallel.start_adjoint(ctx.x_all, gx)
                gw = x_all.t() @ go                       # runs while the reduce-scatter is on the links
                return gx, gw
        before = parallel.ADJOINT_EARLY_STARTS
        xe = x[lo:lo + dg.n].clone().requires_grad_(True)
        we = ic.features(6, 16, 8).requires_grad_(True)
        xa, _ = parallel.exchange(xe, dg, edge_only=False, pipelined=True)
        parallel.finish(xa)
        co = torch.sin(torch.arange(n * 8, dtype=torch.float32).view(n, 8)) * (rank + 1)
        (Consumer.apply(xa, we) * co).sum().backward()
        assert parallel.ADJOINT_EARLY_STARTS == before + 1
        assert parallel.ADJOINT_EARLY_TAKEN == parallel.ADJOINT_EARLY_STARTS and not parallel._ADJOINT_OPEN      # started == picked up
        co_tot = torch.sin(torch.arange(n * 8, dtype=torch.float32).view(n, 8)) * tot
        assert torch.allclose(xe.grad, (co_tot @ we.detach().t())[lo:lo + dg.n], atol=1e-4 * world)
        assert torch.allclose(we.grad, x.t() @ co, atol=1e-3)
        assert not parallel.start_adjoint(x, x)           # not a gathered table: no-op
        # two consumers of one table: autograd sums their gradients into a new tensor, the early result is for one of them
        # only and must NOT be taken - the plain path runs (and is correct)
        xe2 = x[lo:lo + dg.n].clone().requires_grad_(True)
        xa2, _ = parallel.exchange(xe2, dg, edge_only=False)
        ((Consumer.apply(xa2, we.detach()) * co).sum() + (xa2 * w * (rank + 1)).sum()).backward()
        assert torch.allclose(xe2.grad, (co_tot @ we.detach().t() + w * tot)[lo:lo + dg.n], atol=1e-4 * world)

        # halo exchange + its fixed-order adjoint
        os.environ["DISGAT_EXCHANGE"] = "halo"
        xh = x[lo:lo + dg.n].clone().requires_grad_(True)
        x_ref, gc = parallel.exchange(xh, dg, edge_only=True)
        plan = dg._halo
        assert torch.equal(x_ref.detach(), x[plan.ref]) and torch.equal(plan.ref[gc.col.long()], dg.col.long())
        (x_ref * w[plan.ref] * (rank + 1)).sum().backward()
        refd = torch.zeros(n)
        refd[plan.ref] = 1.0
        cnt = refd * (rank + 1)
        dist.all_reduce(cnt)
        assert torch.allclose(xh.grad, (torch.arange(n, dtype=torch.float32) * cnt).unsqueeze(1).expand(n, 16)[lo:lo + dg.n])
        os.environ.pop("DISGAT_EXCHANGE")

        # gradient bucket: a parameter only the even ranks have a gradient for, one nobody has, one everybody has
        lin = torch.nn.Linear(3, 2)
        extra = torch.nn.Linear(2, 2)
        with torch.no_grad():
            for prm in list(lin.parameters()) + list(extra.parameters()):
                prm.fill_(0.5)
        lin.weight.grad = torch.full_like(lin.weight, float(rank + 1))
        lin.bias.grad = torch.full_like(lin.bias, 2.0) if rank % 2 == 0 else None
        parallel.all_reduce_grads([lin, extra], dg)
        assert torch.equal(lin.weight.grad, torch.full_like(lin.weight, tot))
        assert torch.equal(lin.bias.grad, torch.full_like(lin.bias, 2.0 * ((world + 1) // 2)))
        assert extra.weight.grad is None and extra.bias.grad is None
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def _run_ranks(target, world, args=(), timeout=240):
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
            if res[-1][1] != "ok":
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


def test_row_sharding_world2():
    _run_ranks(_worker, 2)


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 4, 8])
def test_exchange_paths_at_baseline_world_sizes(world):
    _run_ranks(_worker_many, world)


def test_balanced_ranges_skewed():
    from edgedisentangle_ssl_amd.parallel import balanced_row_ranges
    deg = torch.tensor([1000] + [1] * 999)
    rp = torch.zeros(1001, dtype=torch.int64)
    rp[1:] = torch.cumsum(deg, 0)
    b = balanced_row_ranges(rp, 4)
    assert b[0] == 0 and b[-1] == 1000 and torch.all(b[1:] >= b[:-1])
    per = [int(rp[b[i + 1]] - rp[b[i]]) for i in range(4)]
    assert max(per) <= 1000 + 2 and sum(per) == int(rp[-1])
