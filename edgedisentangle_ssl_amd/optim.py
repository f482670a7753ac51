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
