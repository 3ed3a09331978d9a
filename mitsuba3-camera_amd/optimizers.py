"""mi.ad.Adam / mi.ad.SGD on torch tensors (src/python/python/ad/optimizers.py:116-302).

Usage mirrors the reference:
    opt = mi.Adam(lr=0.05); opt[key] = params[key]; params.update(opt)
    loop: img = mi.render(scene, params, spp=..., seed=it); loss.backward(); opt.step(); params.update(opt)
Variables are leaf tensors with requires_grad=True living on the scene's device; `step()` reads their .grad
(filled by the render op's backward) and replaces them with fresh leaves, like the reference re-enables gradients.
"""
import math
from collections import defaultdict


class Optimizer:
    def __init__(self, lr, params=None):
        self.lr_default = float(lr)
        self.variables = {}
        self.state = {}
        self.lr = {}
        if params is not None:
            for k in params.keys():
                self[k] = params[k]

    def __contains__(self, k): return k in self.variables
    def __getitem__(self, k): return self.variables[k]
    def keys(self): return self.variables.keys()
    def items(self): return self.variables.items()

    def __setitem__(self, key, value):
        v = value.detach().clone().requires_grad_(True)
        if key not in self.variables or self.variables[key].shape != v.shape:
            self.reset_state = True
            self.state.pop(key, None)
        self.variables[key] = v

    def set_learning_rate(self, lr):
        if isinstance(lr, dict):
            self.lr.update({k: float(v) for k, v in lr.items()})
        else:
            self.lr_default = float(lr)

    def _lr(self, k):
        return self.lr.get(k, self.lr_default)


class SGD(Optimizer):
    """optimizers.py:116-201 (momentum optional)"""

    def __init__(self, lr, momentum=0.0, params=None):
        assert 0 <= momentum < 1 and lr > 0
        self.momentum = momentum
        super().__init__(lr, params)

    def step(self):
        import torch
        for k, p in list(self.variables.items()):
            g = p.grad
            if g is None:
                continue
            if self.momentum == 0:
                u = p.detach() - self._lr(k) * g
            else:
                v = self.state.get(k)
                v = self.momentum * v + g if v is not None else g.clone()
                self.state[k] = v
                u = p.detach() - self._lr(k) * v
            self.variables[k] = u.requires_grad_(True)


class Adam(Optimizer):
    """optimizers.py:204-302, including mask_updates and the UniformAdam variant."""

    def __init__(self, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-8, mask_updates=False, uniform=False, params=None):
        assert 0 <= beta_1 < 1 and 0 <= beta_2 < 1 and lr > 0 and epsilon > 0
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.mask_updates, self.uniform = mask_updates, uniform
        self.t = defaultdict(int)
        super().__init__(lr, params)

    def step(self):
        import torch
        for k, p in list(self.variables.items()):
            g = p.grad
            if g is None:
                continue
            self.t[k] += 1
            lr_t = self._lr(k) * math.sqrt(1 - self.beta_2 ** self.t[k]) / (1 - self.beta_1 ** self.t[k])
            st = self.state.get(k)
            if st is None or st[0].shape != g.shape:
                st = (torch.zeros_like(g), torch.zeros_like(g))
            m_tp, v_tp = st
            m_t = self.beta_1 * m_tp + (1 - self.beta_1) * g
            v_t = self.beta_2 * v_tp + (1 - self.beta_2) * g * g
            if self.mask_updates:
                nz = g != 0
                m_t = torch.where(nz, m_t, m_tp); v_t = torch.where(nz, v_t, v_tp)
            self.state[k] = (m_t, v_t)
            denom = (torch.sqrt(v_t.max()) if self.uniform else torch.sqrt(v_t)) + self.epsilon
            step = lr_t * m_t / denom
            if self.mask_updates:
                step = torch.where(nz, step, torch.zeros_like(step))
            self.variables[k] = (p.detach() - step).requires_grad_(True)
