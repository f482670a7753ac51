"""gloo all_reduce(MAX) on a CUDA scalar produced by a kernel just before (two ranks on one GPU)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def w(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from edgedisentangle_ssl_amd import ops_gemm
    torch.cuda.set_device(0)
    for it in range(5):
        x = torch.randn(200000, 64, device="cuda") * (1.0 + 100.0 * rank)
        am = ops_gemm.amax(x)
        loc = float(am)
        am2 = ops_gemm.amax(x)
        dist.all_reduce(am2, op=dist.ReduceOp.MAX)
        y = am2 * 1.0
        print(rank, it, "local", loc, "reduced", float(am2), float(y), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(w, args=(29533,), nprocs=2)
