"""The collective wrappers of parallel.py over REAL RCCL (backend "nccl"), with the one GPU a test box has: a one-rank
process group.  It cannot show scaling, but it does run every collective form the sharded path issues - all-gather
(tensor and list forms), reduce-scatter (the all-gather's adjoint), all-reduce (f64 sums, MAX, 0-d), all-to-all with
split sizes on int64 and f32 (halo exchange), barrier - through the package's own code on device tensors, so a wrong
dtype / contiguity / split argument fails here and not first on the 8-GPU node.  Runs in a child process (a process
group is per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, socket
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from edgedisentangle_ssl_amd import parallel
from edgedisentangle_ssl_amd.graph import CSRGraph
import inputs_common as ic

idx, _v, n = ic.tiny_graph()
g = CSRGraph.from_index(idx.to(dev), n)
dg = parallel.DistGraph(g.n, g.rowptr, g.col, g.row, g.n, 0, [g.n])          # one rank owning every row
x = ic.features(3, n, 16).to(dev)

# all-gather of rows + its reduce-scatter adjoint (equal counts) through the autograd wrapper
xl = x.clone().requires_grad_(True)
xa = parallel._gather_rows(xl, dg.counts, None, 1, False)
assert torch.equal(xa.detach(), x)
w = torch.arange(n, dtype=torch.float32, device=dev).unsqueeze(1)
(xa * w).sum().backward()
assert torch.allclose(xl.grad, w.expand(n, 16))
# the pipelined form: asynchronous slices left pending, consumed by the column-side GEMM as they land, then the
# reduce-scatter adjoint; the operand bound's MAX-reduce on the fp32 device scalar the GEMMs read
from edgedisentangle_ssl_amd import ops_gemm
xw = ic.features(5, n, 64).to(dev)
w64 = (ic.features(6, 64, 256).to(dev) * 0.1).requires_grad_(True)
xp = xw.clone().requires_grad_(True)
xg = parallel._gather_rows(xp, dg.counts, None, 3, True)
pend = parallel.pending_of(xg)
assert pend is not None and len(pend.slices) == 3
amax = parallel.all_reduce_max(ops_gemm.amax(xp.detach()), dg)
q = parallel.project_gathered(xg, w64, amax)
assert parallel.pending_of(xg) is None
ref_q = xw.double() @ w64.detach().double()
assert float((q.detach().double() - ref_q).abs().max()) <= 2e-6 * float(ref_q.abs().max())
q.sum().backward()
assert torch.allclose(xp.grad, w64.detach().sum(1).expand(n, 64), rtol=1e-5, atol=1e-5)
xg2 = parallel._gather_rows(xw, dg.counts, None, 4, True)
assert torch.equal(parallel.finish(xg2), xw) and parallel.pending_of(xg2) is None
# the list form of all-gather (what ragged, nnz-balanced ranges use)
bufs = [torch.empty_like(x)]
dist.all_gather(bufs, x)
assert torch.equal(bufs[0], x)

# halo exchange: plan (int64 all-to-all with splits), forward rows, adjoint
plan = parallel.HaloPlan(dg)
ref = torch.unique(g.col.long())
assert torch.equal(plan.ref, ref) and plan.send_counts == [ref.numel()] and plan.recv_counts == [ref.numel()]
xh = x.clone().requires_grad_(True)
xr = parallel._HaloRows.apply(xh, plan)
assert torch.equal(xr.detach(), x[ref]) and torch.equal(plan.ref[plan.graph_c.col.long()], g.col.long())
xr.sum().backward()
want = torch.zeros(n, device=dev); want[ref] = 1.0
assert torch.equal(xh.grad, want.unsqueeze(1).expand(n, 16))

# loss / gradient reductions as the trainers issue them
acc = torch.tensor([1.5, 2.5, 3.0, 4.0], dtype=torch.float64, device=dev)
dist.all_reduce(acc); assert acc.tolist() == [1.5, 2.5, 3.0, 4.0]
s0 = torch.tensor(2.0, device=dev); dist.all_reduce(s0); assert float(s0) == 2.0
cnt = torch.tensor([7], dtype=torch.int64, device=dev); dist.all_reduce(cnt); assert int(cnt) == 7
mx = torch.tensor([0.25], dtype=torch.float64, device=dev); dist.all_reduce(mx, op=dist.ReduceOp.MAX); assert float(mx) == 0.25
flat = torch.cat([torch.ones(12, device=dev), torch.zeros(3, device=dev), torch.tensor([1.0, 0.0], device=dev)])   # a gradient bucket + has-grad flags
dist.all_reduce(flat); assert float(flat.sum()) == 13.0
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL-OK")
'''


def test_parallel_wrappers_over_rccl_one_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
