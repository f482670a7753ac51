import sys, torch, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from edgedisentangle_ssl_amd import ops_gemm
from kbench import timeit
dev = torch.device("cuda")
for M in (8_000_000, 1_000_000):
    x = torch.randn(M, 256, device=dev)
    lin = torch.nn.Linear(256, 8).to(dev)
    with torch.no_grad():
        a = timeit(lambda: lin(x), 5)
        b = timeit(lambda: ops_gemm.skinny_linear(x, lin), 5)
        ls = lambda t: torch.log_softmax(t, dim=1)
        y = lin(x)
        c = timeit(lambda: ls(y), 5)
    print(M, "F.linear %.3f ms   skinny %.3f ms  (%.2f TB/s)   log_softmax %.3f" % (a, b, M * 1024 / b / 1e9, c))
