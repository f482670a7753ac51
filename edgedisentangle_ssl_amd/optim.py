"""The trainers' optimisers: one Adam per sub-module as in the reference (trainer.py:58-60 - the encoder, each fuser
and each classifier get their own torch.optim.Adam(lr, weight_decay), and every train_step steps all of them,
trainer.py:205-206, pretrainer.py:754-756), stepped together in ONE HIP launch (csrc/optim.hip) instead of 3-5
optimiser steps of a few small tensors each.  Semantics are torch.optim.Adam's: L2 weight decay folded into the
gradient, bias-corrected moments, independent state per optimiser; a parameter whose gradient is None is skipped and
its step count does not advance.  Step counts live on the host (whether a gradient exists is a host fact), so a step
costs no device synchronisation.
"""
import ctypes
import math

import torch

from . import _lib


class ModuleAdam:
    """Adam state of one sub-module's parameters (what one entry of the reference's `models_opt` holds)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.state = {}                        # id(param) -> [step, exp_avg, exp_avg_sq]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_().zero_()

    def _entries(self):
        """(param, grad, exp_avg, exp_avg_sq, step_size, inv_sqrt_bc2, wd) of every parameter that has a gradient;
        advances those parameters' step counts."""
        b1, b2 = self.betas
        out = []
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("MultiAdam: parameters must be contiguous fp32 device tensors (no CPU fallback)")
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            st = self.state.get(id(p))
            if st is None:
                st = self.state[id(p)] = [0, torch.zeros_like(p), torch.zeros_like(p)]
            st[0] += 1
            t = st[0]
            out.append((p, g, st[1], st[2], self.lr / (1.0 - b1 ** t), 1.0 / math.sqrt(1.0 - b2 ** t), self.weight_decay))
        return out

    def step(self):
        step_all([self])

    def state_dict(self):
        return {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                "state": [None if id(p) not in self.state else
                          {"step": self.state[id(p)][0], "exp_avg": self.state[id(p)][1], "exp_avg_sq": self.state[id(p)][2]}
                          for p in self.params]}

    def load_state_dict(self, sd):
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
        self.state = {}
        for p, st in zip(self.params, sd["state"]):
            if st is not None:
                self.state[id(p)] = [int(st["step"]), st["exp_avg"].to(p.device).clone(), st["exp_avg_sq"].to(p.device).clone()]


def step_all(optimisers):
    """One Adam step of every optimiser in `optimisers` - all their tensors in one launch per 64 tensors.  Optimisers
    with different betas / eps are launched separately (the reference never mixes them)."""
    groups = {}
    for o in optimisers:
        groups.setdefault((o.betas, o.eps), []).extend(o._entries())
    stream = torch.cuda.current_stream().cuda_stream
    for ((b1, b2), eps), ent in groups.items():
        n = len(ent)
        if n == 0:
            continue
        ptr = lambda k: (ctypes.c_void_p * n)(*[e[k].data_ptr() for e in ent])           # noqa: E731
        flt = lambda k: (ctypes.c_float * n)(*[e[k] for e in ent])                        # noqa: E731
        numel = (ctypes.c_int64 * n)(*[e[0].numel() for e in ent])
        with torch.no_grad():
            _lib.call("disgat_adam_multi", n, ptr(0), ptr(1), ptr(2), ptr(3), numel, flt(4), flt(5), flt(6),
                      b1, b2, eps, stream)
        # the kernel wrote the parameters through raw pointers: advance autograd's version counters as an in-place
        # torch op would (saved-tensor checks and the packed-weight memo in layers._memo key on them)
        torch.autograd.graph.increment_version([e[0] for e in ent])


class DeviceStepAdam:
    """step_all() for a train_step that replays from a HIP graph: the step counts live in a device int32 tensor the step
    advances itself and the kernel forms the bias corrections from them (disgat_adam_multi_dev) - the host-side
    corrections of step_all() would be frozen into the graph at their capture-time values.  The host counts of the
    ModuleAdam states stay the truth between modes: `sync()` loads them into the device tensor when they moved without
    it (eager steps, a rolled-back warm-up, load_state_dict), `advance_host()` is called once per executed / replayed step."""

    def __init__(self, optimisers):
        self.optimisers = list(optimisers)
        self.keys = None            # (optimiser index, param index) of every parameter that receives a gradient
        self.steps = None
        self.mirror = None

    def _entries(self):
        ent, keys = [], []
        for oi, o in enumerate(self.optimisers):
            for pi, p in enumerate(o.params):
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("DeviceStepAdam: parameters must be contiguous fp32 device tensors (no CPU fallback)")
                if not g.is_contiguous():
                    g = p.grad = g.contiguous()
                st = o.state.get(id(p))
                if st is None:
                    st = o.state[id(p)] = [0, torch.zeros_like(p), torch.zeros_like(p)]
                ent.append((p, g, st[1], st[2], o.lr, o.weight_decay, o, st))
                keys.append((oi, pi))
        return ent, keys

    def host_steps(self, ent):
        return [e[7][0] for e in ent]

    def step(self):
        """Enqueue: steps += 1, then one launch per (betas, eps) group.  No host state changes (see advance_host)."""
        ent, keys = self._entries()
        if self.keys is None:
            self.keys = keys
            dev = ent[0][0].device if ent else torch.device("cuda")
            self.steps = torch.tensor(self.host_steps(ent), dtype=torch.int32, device=dev)
            self.mirror = self.host_steps(ent)
        elif keys != self.keys:
            raise RuntimeError("DeviceStepAdam: the set of parameters with a gradient changed between steps of a captured trainer")
        self._last = ent
        if not ent:
            return
        self.steps.add_(1)
        stream = torch.cuda.current_stream().cuda_stream
        order = sorted(range(len(ent)), key=lambda i: (ent[i][6].betas, ent[i][6].eps))
        if order != list(range(len(ent))):
            raise RuntimeError("DeviceStepAdam: optimisers with different betas / eps must be listed in groups")
        i = 0
        while i < len(ent):
            o0 = ent[i][6]
            j = i
            while j < len(ent) and (ent[j][6].betas, ent[j][6].eps) == (o0.betas, o0.eps):
                j += 1
            grp, n = ent[i:j], j - i
            ptr = lambda k: (ctypes.c_void_p * n)(*[e[k].data_ptr() for e in grp])          # noqa: E731
            numel = (ctypes.c_int64 * n)(*[e[0].numel() for e in grp])
            lr = (ctypes.c_double * n)(*[e[4] for e in grp])
            wd = (ctypes.c_float * n)(*[e[5] for e in grp])
            with torch.no_grad():
                _lib.call("disgat_adam_multi_dev", n, ptr(0), ptr(1), ptr(2), ptr(3), numel, lr, wd,
                          self.steps.data_ptr() + 4 * i, o0.betas[0], o0.betas[1], o0.eps, stream)
            i = j

    def sync(self):
        """Before a step runs or replays: make the device counts equal the host's if those moved on their own."""
        if self.keys is None:
            return
        host = [self.optimisers[oi].state[id(self.optimisers[oi].params[pi])][0] for oi, pi in self.keys]
        if host != self.mirror:
            self.steps.copy_(torch.tensor(host, dtype=torch.int32))
            self.mirror = host

    def advance_host(self):
        """After a step ran (eagerly or as a replay): the host counts follow, parameter versions advance."""
        params = []
        for oi, pi in self.keys or []:
            o = self.optimisers[oi]
            p = o.params[pi]
            o.state[id(p)][0] += 1
            params.append(p)
        self.mirror = [m + 1 for m in self.mirror] if self.mirror is not None else None
        if params:
            torch.autograd.graph.increment_version(params)
