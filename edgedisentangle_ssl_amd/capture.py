"""HIP-graph capture of the (launch-bound) small-graph path.

On Cora / chameleon-sized graphs a DISGAT forward is ~100 kernel launches of a few microseconds
each, so host launch overhead dominates (get_em 0.85 ms eager).  Every launcher of libdisgat_hip.so
is stream-ordered and allocation-free, so a whole forward (or forward + SSL losses on resident pair
lists) can be captured once and replayed as one graph launch (0.32 ms on Cora).  Shapes must stay
fixed between replays; inputs are updated in place through the static tensors.
"""
import torch


class CapturedCall:
    """`fn(*static_inputs)` captured into one HIP graph.  call(*new_inputs) copies the new values
    into the static input buffers, replays, and returns the (static) outputs."""

    def __init__(self, fn, *static_inputs, warmup=3):
        self.static_inputs = static_inputs
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                      # builds the CSR cache, hipBLASLt workspaces, ...
                fn(*static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.outputs = fn(*static_inputs)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            if src is not dst and torch.is_tensor(dst) and not dst.is_sparse:
                dst.copy_(src)
        self.graph.replay()
        return self.outputs


def capture_get_em(encoder, x, adj, fusers):
    """One-graph replay of encoder.get_em(x, adj, fusers) for fixed shapes (inference)."""
    from .graph import graph_of
    g = graph_of(adj)
    encoder.eval()
    return CapturedCall(lambda feats: encoder.get_em(feats, g, fusers), x)
