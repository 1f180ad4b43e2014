"""Masked Adam (neural_renderer/optimizers.py:9-39): skip elements whose gradient is exactly zero, honour a
per-parameter `lr` multiplier.  Only neural_renderer's own examples use it; 3D-SDN trains with torch.optim.Adam
(geometric/scripts/main.py:188,439)."""
import math

import torch


class Adam(torch.optim.Optimizer):
    def __init__(self, params, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        super(Adam, self).__init__(params, dict(lr=alpha, beta1=beta1, beta2=beta2, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st['m'] = torch.zeros_like(p)
                    st['v'] = torch.zeros_like(p)
                    st['t'] = 0
                # chainer's UpdateRule.update counts the step before update_core, and AdamRule.lr is the bias-corrected
                # step size alpha * sqrt(1 - beta2^t) / (1 - beta1^t) (chainer 4.1.0 optimizers/adam.py)
                st['t'] += 1
                t = st['t']
                lr_t = group['lr'] * math.sqrt(1.0 - group['beta2'] ** t) / (1.0 - group['beta1'] ** t)
                lr = lr_t * getattr(p, 'lr', 1.0)
                if lr == 0:
                    continue
                g, m, v = p.grad, st['m'], st['v']
                mask = g != 0
                m_new = m + (1 - group['beta1']) * (g - m)
                v_new = (v + (1 - group['beta2']) * (g * g - v)).clamp_(min=0)
                m.copy_(torch.where(mask, m_new, m))
                v.copy_(torch.where(mask, v_new, v))
                p.sub_(torch.where(mask, lr * m / (v.sqrt() + group['eps']), torch.zeros_like(p)))
        return loss
