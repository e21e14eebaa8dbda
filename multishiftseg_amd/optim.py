"""Adam with L2-coupled weight decay as one HIP kernel per parameter (mss_adam_step_f32):
the arithmetic of torch.optim.Adam(params, lr, weight_decay) that train_deeplab.py:134-149 builds
(not AdamW; rebuilt from scratch, state included, at the stage switch train_deeplab.py:151-166)."""
import torch

from ._lib import call, ptr


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.state = {}
        self.step_count = 0

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.get(id(p))
            if st is None:
                st = self.state[id(p)] = (torch.zeros_like(p), torch.zeros_like(p))
            g = p.grad.contiguous()
            call("mss_adam_step_f32", ptr(p), ptr(g), ptr(st[0]), ptr(st[1]), p.numel(), float(self.lr),
                 float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                 self.step_count)
            torch.autograd.graph.increment_version(p)   # packed-weight caches key on tensor._version
