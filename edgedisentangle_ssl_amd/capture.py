"""HIP-graph capture of the (launch-bound) small-graph path.

On Cora / chameleon-sized graphs a DISGAT forward is ~100 kernel launches of a few microseconds
each, so host launch overhead dominates (get_em 0.85 ms eager).  Every launcher of libdisgat_hip.so
is stream-ordered and allocation-free, so a whole forward (or forward + SSL losses on resident pair
lists) can be captured once and replayed as one graph launch (0.32 ms on Cora).  Shapes must stay
fixed between replays; inputs are updated in place through the static tensors.
"""
import torch


class _no_gc:
    """No cyclic garbage collection while a stream is capturing.  A collection that happens to run inside the capture
    can finalise objects of EARLIER work - a HIP graph, events, a trainer with its StaticStep (they form a reference
    cycle, so only the cyclic collector frees them) - and destroying those is not permitted during capture: the process
    aborts inside the destructor.  torch.cuda.graph() collects once on entry; this keeps the collector off until exit."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


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
        with torch.no_grad(), _no_gc(), torch.cuda.graph(self.graph):
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


# ---------------------------------------------------------------------------------------------------------------
# Whole train_steps as HIP graphs.  On Cora / chameleon a train_step is ~400 launches of a few microseconds each
# behind ~7 ms of Python + autograd-engine work; replayed from a graph, the host does two things per step: the replay
# call and the bookkeeping of step counts (the pair samplers' generator state lives on the device: csrc/pair_sample.hip).
# What makes a step replayable:
#   * pair lists of FIXED capacity, sampled on the device with their valid length on the device
#     (sampling.PairSampler.sample_static; padding carries label -1, which disgat_pair_loss / _bwd skip);
#   * backward segment structures built without reading sizes back (ops_bwd._segments_static);
#   * attention-dropout seeds = a per-call-site constant + a device counter the step advances (layers.StepSeed);
#   * Adam's step counts on the device (optim.DeviceStepAdam / disgat_adam_multi_dev).
# The first call runs the step eagerly once to build every cache (CSR transpose, work items, hipBLASLt
# workspaces), ROLLS the trainer BACK to where it started (parameters, Adam moments and counts, seed counter, RNG
# streams), captures, and from then on replays - so step 1 of the run already comes from the graph and a captured
# run follows the same trajectory as the same static step run eagerly (tests/test_gpu_capture.py).
class StaticStep:
    """Owned by its trainer (Trainer.static_step keeps it); holds the trainer only WEAKLY and takes the trainer's step
    functions unbound, so trainer and step form no reference cycle: dropping the trainer (or Trainer.close()) frees the step
    and its HIP graph right there, by reference count - not whenever the cyclic collector next runs, which may be in the middle
    of somebody else's stream capture, where destroying a graph aborts the process."""

    def __init__(self, trainer, host_fn, device_fn, warmup=1):
        from . import layers, optim
        import os
        import weakref
        self._trainer = weakref.ref(trainer)
        self._host_fn = getattr(host_fn, "__func__", host_fn)          # unbound: a bound method would hold the trainer
        self._device_fn = getattr(device_fn, "__func__", device_fn)
        self.warmup = int(os.environ.get("DISGAT_CAPTURE_WARMUP", warmup))
        dev = next(trainer.models[0].parameters()).device
        self.seed = layers.StepSeed(dev)
        self.adam = optim.DeviceStepAdam(trainer.models_opt)
        self.graph = None
        self.logs = None
        self.replays = 0

    @property
    def trainer(self):
        tr = self._trainer()
        if tr is None:
            raise RuntimeError("StaticStep: its trainer is gone")
        return tr

    def host_fn(self, *args):
        """Host half of a step: the trainer's own (nothing, today) + the pair samplers' seeds.  A sampler takes its seed from
        torch's CPU generator at first use - a device fill that must not end up inside the graph (every replay would reset
        the generator's step with it)."""
        tr = self.trainer
        self._host_fn(tr, *args)
        for smp in tr.samplers(*args):
            if hasattr(smp, "_ensure_seed"):
                smp._ensure_seed()

    def device_fn(self, adam, *args):
        return self._device_fn(self.trainer, adam, *args)

    def close(self):
        """Free the HIP graph now."""
        self.graph = None
        self.logs = None

    # -- one execution of the step's device half (eager or under capture)
    def _body(self, *args):
        from . import layers
        tr = self.trainer
        tr._begin_step()
        prev, layers.STEP_SEED = layers.STEP_SEED, self.seed
        try:
            self.seed.begin_step()
            return self.device_fn(self.adam, *args)
        finally:
            layers.STEP_SEED = prev

    def run_eager(self, *args):
        """The static step without a graph (reference behaviour of the captured one; also the warm-up)."""
        self.host_fn(*args)
        self.adam.sync()
        logs = self._body(*args)
        self.adam.advance_host()
        return logs

    def _snapshot(self, *args):
        tr = self.trainer
        samplers = [(smp, smp._seeded, smp.meta.clone()) for smp in tr.samplers(*args) if hasattr(smp, "meta")]
        params = [p for m in tr.models for p in m.parameters()]
        opt = []
        for o in tr.models_opt:
            opt.append({id(p): (st[0], st[1].clone(), st[2].clone()) for p in o.params for st in [o.state.get(id(p))] if st is not None})
        return dict(params=[(p, p.detach().clone()) for p in params], opt=opt, seed=self.seed.counter.clone(),
                    cpu_rng=torch.get_rng_state(), cuda_rng=torch.cuda.get_rng_state(), samplers=samplers)

    def _restore(self, snap):
        tr = self.trainer
        with torch.no_grad():
            for p, v in snap["params"]:
                p.copy_(v)
            for o, saved in zip(tr.models_opt, snap["opt"]):
                for p in o.params:
                    st = o.state.get(id(p))
                    if st is None:
                        continue
                    old = saved.get(id(p))
                    if old is None:                  # created by the warm-up: back to a fresh state, same tensors
                        st[0] = 0
                        st[1].zero_()
                        st[2].zero_()
                    else:
                        st[0] = old[0]
                        st[1].copy_(old[1])
                        st[2].copy_(old[2])
            self.seed.counter.copy_(snap["seed"])
            for smp, seeded, meta in snap["samplers"]:      # generator (seed, step) and event counters back; an unseeded sampler
                smp.meta.copy_(meta)                        # draws its seed again, from the restored CPU generator
                smp._seeded = seeded
        torch.set_rng_state(snap["cpu_rng"])
        torch.cuda.set_rng_state(snap["cuda_rng"])

    def _capture(self, *args):
        snap = self._snapshot(*args)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.run_eager(*args)
        torch.cuda.current_stream().wait_stream(side)
        self._restore(snap)
        self.host_fn(*args)             # the capture pass itself consumes no host randomness that a replay would not
        self.adam.sync()
        for o in self.trainer.models_opt:
            o.zero_grad()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()          # (a capture the stream refuses - a host read inside the step, say - raises here;
        with _no_gc(), torch.cuda.graph(graph):     # on this ROCm the process does not recover from that: fix the step or
            self.logs = self._body(*args)           # run with --capture off)
        self.graph = graph
        self._restore(snap)             # the host generator again: the first replay draws what an uncaptured first step would

    def __call__(self, *args):
        if self.graph is None:
            self._capture(*args)
        self.host_fn(*args)
        self.adam.sync()
        self.graph.replay()
        self.adam.advance_host()
        self.replays += 1
        return self.logs
