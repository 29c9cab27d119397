"""Flat-buffer Adam: all parameters (and their gradients) of a model are views into ONE contiguous
fp32 buffer, so the optimiser step is one esc_adam_step launch and the data-parallel gradient
exchange is one RCCL all-reduce of `flat_grad`.  Arithmetic = torch.optim.Adam
(reference run_graphcount.py:478: Adam(model.parameters(), lr), defaults betas=(0.9,0.999), eps=1e-8).
"""
import torch

from . import _native as nv


class FlatAdam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam: no parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam runs on the HIP device only; there is no CPU fallback")
        n = sum(p.numel() for p in self.params)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + k].view(p.shape)
            p.grad = self.flat_grad[off:off + k].view(p.shape)
            off += k
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, params=self.params)]
        self.step_count = 0

    def zero_grad(self, set_to_none=False):
        """Gradients are persistent views of flat_grad: zero in place (autograd then accumulates into them)."""
        self.flat_grad.zero_()
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() < self.flat_grad.data_ptr():
                self._rebind()
                break

    def _rebind(self):
        off = 0
        for p in self.params:
            k = p.numel()
            p.grad = self.flat_grad[off:off + k].view(p.shape)
            off += k

    def step(self):
        g = self.param_groups[0]
        self.step_count += 1
        nv.call("esc_adam_step", nv.ptr(self.flat_param), nv.ptr(self.flat_grad), nv.ptr(self.exp_avg),
                nv.ptr(self.exp_avg_sq), self.flat_param.numel(), float(g["lr"]), float(g["betas"][0]),
                float(g["betas"][1]), float(g["eps"]), self.step_count, nv.stream())

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq,
                    param_groups=[{k: v for k, v in g.items() if k != "params"} for g in self.param_groups])

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
