"""Fused optimizers on the HIP kernels: Adam (laed_adam_step; defaults of torch.optim.Adam as the reference uses it,
laed/models/model_bases.py:122-125: lr = init_lr, betas (0.9, 0.999), eps 1e-8, no weight decay, no clipping), and the
SGD / RMSprop branches of the same function (:126-133)."""
import torch

from . import ops


class FusedAdam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, reducer=None):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.state = {}
        self.reducer = reducer

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        if self.reducer is not None:
            self.reducer.wait()
        by_step = {}                    # torch's Adam counts steps per parameter: one launch group per count
        for p in self.params:
            if p.grad is None:          # e.g. log_py without train_prior: torch's Adam skips grad-less params
                continue
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)}
            st["step"] += 1
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            grp = by_step.setdefault(st["step"], ([], [], [], []))
            for lst, t in zip(grp, (p.data, g, st["m"], st["v"])):
                lst.append(t)
        for step, (ps, gs, ms, vs) in by_step.items():
            ops.adam_step_multi(ps, gs, ms, vs, step, self.lr, self.betas[0], self.betas[1], self.eps)

    def state_dict(self):
        return {"lr": self.lr, "betas": self.betas, "eps": self.eps,
                "state": [self.state.get(p) for p in self.params]}


class _FusedBase(object):
    def __init__(self, params, lr, reducer=None):
        self.params = list(params)
        self.lr = float(lr)
        self.state = {}
        self.reducer = reducer

    zero_grad = FusedAdam.zero_grad

    def _grads(self):
        if self.reducer is not None:
            self.reducer.wait()
        for p in self.params:
            if p.grad is not None:      # torch's optimizers skip grad-less params
                yield p, (p.grad if p.grad.is_contiguous() else p.grad.contiguous())


class FusedSGD(_FusedBase):
    """torch.optim.SGD(params, lr, momentum) (model_bases.py:126-129)."""

    def __init__(self, params, lr, momentum=0.0, reducer=None):
        super(FusedSGD, self).__init__(params, lr, reducer)
        self.momentum = float(momentum)

    @torch.no_grad()
    def step(self):
        for p, g in self._grads():
            buf = None
            if self.momentum != 0.0:
                buf = self.state.get(p)
                if buf is None:
                    buf = self.state[p] = torch.zeros_like(p)
            ops.check(ops.lib.laed_sgd_step(ops.ptr(p.data), ops.ptr(g), ops.ptr(buf), p.numel(), self.lr, self.momentum, 1.0,
                                            ops.stream()), "laed_sgd_step")


class FusedRMSprop(_FusedBase):
    """torch.optim.RMSprop(params, lr, momentum) with torch's alpha = 0.99, eps = 1e-8 (model_bases.py:130-133)."""

    def __init__(self, params, lr, momentum=0.0, alpha=0.99, eps=1e-8, reducer=None):
        super(FusedRMSprop, self).__init__(params, lr, reducer)
        self.momentum, self.alpha, self.eps = float(momentum), float(alpha), float(eps)

    @torch.no_grad()
    def step(self):
        for p, g in self._grads():
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = {"sq": torch.zeros_like(p),
                                      "buf": torch.zeros_like(p) if self.momentum > 0.0 else None}
            ops.check(ops.lib.laed_rmsprop_step(ops.ptr(p.data), ops.ptr(g), ops.ptr(st["sq"]), ops.ptr(st["buf"]), p.numel(),
                                                self.lr, self.alpha, self.eps, self.momentum, 1.0, ops.stream()),
                      "laed_rmsprop_step")
