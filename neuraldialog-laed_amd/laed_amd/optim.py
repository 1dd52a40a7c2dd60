"""Fused Adam on the HIP kernel (laed_adam_step); defaults of torch.optim.Adam as the reference uses it
(laed/models/model_bases.py:122-125: lr = init_lr, betas (0.9, 0.999), eps 1e-8, no weight decay, no clipping)."""
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
        for p in self.params:
            if p.grad is None:          # e.g. log_py without train_prior: torch's Adam skips grad-less params
                continue
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)}
            st["step"] += 1
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            ops.adam_step(p.data, g, st["m"], st["v"], st["step"], self.lr, self.betas[0], self.betas[1], self.eps)

    def state_dict(self):
        return {"lr": self.lr, "betas": self.betas, "eps": self.eps,
                "state": [self.state.get(p) for p in self.params]}
