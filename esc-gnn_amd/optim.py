"""Flat-buffer Adam: all parameters (and their gradients) of a model are views into ONE contiguous
fp32 buffer, so the optimiser step is one esc_adam_step launch and the data-parallel gradient
exchange is one RCCL all-reduce of `flat_grad`.  Arithmetic = torch.optim.Adam
(reference run_graphcount.py:478: Adam(model.parameters(), lr), defaults betas=(0.9,0.999), eps=1e-8).
"""
import torch

from . import _native as nv
from .parallel import FlatBucket


class FlatAdam(FlatBucket):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, late=None):
        params = [p for p in params]
        if params and params[0].device.type != "cuda":
            raise RuntimeError("FlatAdam runs on the HIP device only; there is no CPU fallback")
        super().__init__(params, late=late)
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, params=self.params)]
        self.step_count = 0

    def step(self, grad_denom=None):
        """One Adam update.  grad_denom: device scalar the gradients are divided by inside the launch (the global
        target count returned by `all_reduce_sum`)."""
        g = self.param_groups[0]
        self.step_count += 1
        nv.call("esc_adam_step_scaled", nv.ptr(self.flat_param), nv.ptr(self.flat_grad), nv.ptr(self.exp_avg),
                nv.ptr(self.exp_avg_sq), self.flat_param.numel(), float(g["lr"]), float(g["betas"][0]),
                float(g["betas"][1]), float(g["eps"]), self.step_count, nv.ptr(grad_denom), nv.stream())

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq,
                    param_groups=[{k: v for k, v in g.items() if k != "params"} for g in self.param_groups])

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


class ReduceLROnPlateau(object):
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', threshold=1e-4 rel) as configured at
    reference run_graphcount.py:479-480 (factor=lr_decay_factor, patience, min_lr=1e-5); works on any
    object with `param_groups`."""

    def __init__(self, optimizer, mode="min", factor=0.1, patience=10, min_lr=0.0, threshold=1e-4):
        assert mode == "min"
        self.optimizer, self.factor, self.patience, self.min_lr, self.threshold = optimizer, factor, patience, min_lr, threshold
        self.best, self.num_bad = float("inf"), 0

    def step(self, metric):
        m = float(metric)
        if m < self.best * (1.0 - self.threshold):
            self.best, self.num_bad = m, 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            for g in self.optimizer.param_groups:
                new = max(g["lr"] * self.factor, self.min_lr)
                if g["lr"] - new > 1e-8:
                    g["lr"] = new
            self.num_bad = 0
