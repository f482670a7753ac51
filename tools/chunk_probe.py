import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py", "--fwd-only"]
import bench
from edgedisentangle_ssl_amd import ops
o = bench.parse()
dev = torch.device("cuda")
for chunk in (128, 256, 512, 1024, 4096):
    ops.CHUNK[3] = chunk
    a, enc, trainers, graph, x, lists = bench.build_workload(o, 0, 1, dev)
    for _ in range(2):
        bench.one_step(o, enc, trainers, graph, x, lists)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(8):
        bench.one_step(o, enc, trainers, graph, x, lists)
    torch.cuda.synchronize()
    print(chunk, round((time.perf_counter() - t) / 8 * 1e3, 2), "ms T_fwd", flush=True)
    del graph, x, lists, enc, trainers
    torch.cuda.empty_cache()
