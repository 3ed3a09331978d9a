"""mi.ad.SGD / mi.ad.Adam on torch tensors, statement for statement after src/python/python/ad/optimizers.py:116-302.

Usage mirrors the reference:
    opt = mi.Adam(lr=0.05); opt[key] = params[key]; params.update(opt)
    loop: img = mi.render(scene, params, spp=..., seed=it); loss.backward(); opt.step(); params.update(opt)
Variables are leaf tensors with requires_grad=True living on the scene's device; `step()` reads their .grad
(filled by the render op's backward) and replaces them with fresh leaves, like the reference re-enables gradients.
A variable whose .grad is None is the reference's "gradient of shape 0": no step is taken for it.

Two quirks of the reference are kept on purpose (tests/test_host_cpu.py::test_optimizers_follow_the_reference pins them):
  * SGD with momentum steps with the PREVIOUS velocity (`step = lr * self.state[k]` is formed before the state is replaced,
    optimizers.py:163-170), so the first step of a fresh variable is zero;
  * Adam increments its step counter t[k] BEFORE it looks at the gradient (optimizers.py:265-273), so an iteration without a
    gradient still advances the bias correction; reset(k) (new key, or a shape change) zeroes the moments AND t[k].
"""
import math
from collections import defaultdict


class Optimizer:
    """optimizers.py:6-114"""

    def __init__(self, lr, params=None):
        self.lr = {}
        self.set_learning_rate(lr)
        self.variables = {}
        self.state = {}
        if params is not None:
            for k, v in params.items():
                self.__setitem__(k, v)

    def __contains__(self, key): return key in self.variables
    def __getitem__(self, key): return self.variables[key]
    def __delitem__(self, key): del self.variables[key]
    def __len__(self): return len(self.variables)
    def keys(self): return self.variables.keys()
    def items(self): return self.variables.items()

    def __setitem__(self, key, value):
        """Overwrite the value of a parameter; the optimizer state is kept unless the parameter is new or changed its shape
        (optimizers.py:41-60)."""
        import torch
        if not (torch.is_tensor(value) and value.is_floating_point()):
            raise Exception('Optimizer.__setitem__(): value should be differentiable!')
        needs_reset = (key not in self.variables) or tuple(self.variables[key].shape) != tuple(value.shape)
        self.variables[key] = value.detach().clone().requires_grad_(True)
        if needs_reset:
            self.reset(key)

    def set_learning_rate(self, lr):
        if isinstance(lr, (float, int)):
            self.lr_default = float(lr)
        elif isinstance(lr, dict):
            for k, v in lr.items():
                self.lr[k] = float(v)
        else:
            raise Exception('Optimizer.set_learning_rate(): value should be a float or a dict!')

    def _lr(self, k):
        return self.lr.get(k, self.lr_default)

    def reset(self, key):
        """Resets the internal state associated with a parameter, if any (e.g. momentum)."""
        pass


class SGD(Optimizer):
    """optimizers.py:116-201"""

    def __init__(self, lr, momentum=0, mask_updates=False, params=None):
        assert momentum >= 0 and momentum < 1
        assert lr > 0
        self.momentum = momentum
        self.mask_updates = mask_updates
        super().__init__(lr, params)

    def step(self):
        import torch
        for k, p in list(self.variables.items()):
            g_p = p.grad
            if g_p is None:
                continue
            if self.momentum != 0:
                if tuple(g_p.shape) != tuple(self.state[k].shape):
                    self.reset(k)
                next_state = self.momentum * self.state[k] + g_p
                step = self._lr(k) * self.state[k]                # the previous velocity, as in the reference (:166)
                if self.mask_updates:
                    nonzero = g_p != 0
                    next_state = torch.where(nonzero, next_state, self.state[k])
                    step = torch.where(nonzero, step, torch.zeros_like(step))
                self.state[k] = next_state
                value = p.detach() - step
            else:
                value = p.detach() - self._lr(k) * g_p
            self.variables[k] = value.requires_grad_(True)

    def reset(self, key):
        import torch
        if self.momentum == 0:
            return
        self.state[key] = torch.zeros_like(self.variables[key].detach())


class Adam(Optimizer):
    """optimizers.py:204-302, including mask_updates and the UniformAdam variant."""

    def __init__(self, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-8, mask_updates=False, uniform=False, params=None):
        assert 0 <= beta_1 < 1 and 0 <= beta_2 < 1 and lr > 0 and epsilon > 0
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.mask_updates, self.uniform = mask_updates, uniform
        self.t = defaultdict(lambda: 0)
        super().__init__(lr, params)

    def step(self):
        import torch
        if not self.mask_updates and not self.uniform and len(self.variables) > 4:
            return self._step_fused()
        for k, p in list(self.variables.items()):
            self.t[k] += 1
            lr_scale = math.sqrt(1 - self.beta_2 ** self.t[k]) / (1 - self.beta_1 ** self.t[k])
            lr_t = self._lr(k) * lr_scale
            g_p = p.grad
            if g_p is None:
                continue
            elif tuple(g_p.shape) != tuple(self.state[k][0].shape):
                self.reset(k)
            m_tp, v_tp = self.state[k]
            m_t = self.beta_1 * m_tp + (1 - self.beta_1) * g_p
            v_t = self.beta_2 * v_tp + (1 - self.beta_2) * g_p * g_p
            if self.mask_updates:
                nonzero = g_p != 0
                m_t = torch.where(nonzero, m_t, m_tp); v_t = torch.where(nonzero, v_t, v_tp)
            self.state[k] = (m_t, v_t)
            if self.uniform:
                step = lr_t * m_t / (torch.sqrt(v_t.max()) + self.epsilon)
            else:
                step = lr_t * m_t / (torch.sqrt(v_t) + self.epsilon)
            if self.mask_updates:
                step = torch.where(nonzero, step, torch.zeros_like(step))
            self.variables[k] = (p.detach() - step).requires_grad_(True)

    def _step_fused(self):
        """The same update (optimizers.py:263-302) for ALL variables at once instead of nine launches per variable: a lens design holds ~30
        one-element variables, and 270 tiny launches were 2 of the 30 ms of a design iteration.  On the GPU: ONE launch of `mi355_adam_step`
        (m and v updated in place, the new values written into one fresh buffer the new variables are views of); elsewhere a handful of
        multi-tensor calls."""
        import torch
        ks, ps, gs, ms, vs, lrs = [], [], [], [], [], []
        for k, p in list(self.variables.items()):
            self.t[k] += 1
            g_p = p.grad
            if g_p is None:
                continue
            if tuple(g_p.shape) != tuple(self.state[k][0].shape):
                self.reset(k)
            lr_scale = math.sqrt(1 - self.beta_2 ** self.t[k]) / (1 - self.beta_1 ** self.t[k])
            ks.append(k); ps.append(p.detach()); gs.append(g_p); ms.append(self.state[k][0]); vs.append(self.state[k][1]); lrs.append(self._lr(k) * lr_scale)
        if not ks:
            return
        if all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for group in (ps, gs, ms, vs) for t in group):
            return self._step_kernel(ks, ps, gs, ms, vs, lrs)
        m_t = torch._foreach_mul(ms, self.beta_1); torch._foreach_add_(m_t, torch._foreach_mul(gs, 1 - self.beta_1))
        v_t = torch._foreach_mul(vs, self.beta_2); torch._foreach_add_(v_t, torch._foreach_mul(torch._foreach_mul(gs, 1 - self.beta_2), gs))
        den = torch._foreach_sqrt(v_t); torch._foreach_add_(den, self.epsilon)
        step = torch._foreach_div(torch._foreach_mul(m_t, lrs), den)
        new = torch._foreach_sub(ps, step)
        for k, m, v, n in zip(ks, m_t, v_t, new):
            self.state[k] = (m, v)
            self.variables[k] = n.requires_grad_(True)

    def _step_kernel(self, ks, ps, gs, ms, vs, lrs):
        import ctypes as C
        import torch
        from . import _abi as abi
        dev = ps[0].device
        if any(t.device != dev for group in (ps, gs, ms, vs) for t in group):
            raise RuntimeError("Adam: variables, gradients and state must live on one device")
        counts = [p.numel() for p in ps]
        flat = torch.empty(sum(counts), dtype=torch.float32, device=dev)
        slots = (abi.AdamSlot * len(ks))()
        base = flat.data_ptr(); off = 0; new = []
        for i, (p, g, m, v, lr, n) in enumerate(zip(ps, gs, ms, vs, lrs, counts)):
            if m.data_ptr() == v.data_ptr():                             # (reset() hands out two tensors; be safe against an aliased pair)
                v = v.clone(); self.state[ks[i]] = (m, v)
            sl = slots[i]
            sl.param = p.data_ptr(); sl.grad = g.data_ptr(); sl.m = m.data_ptr(); sl.v = v.data_ptr(); sl.param_out = base + 4 * off; sl.lr_t = lr; sl.count = n
            new.append(flat[off:off + n].view(p.shape)); off += n
        with torch.cuda.device(dev):
            abi.check(abi.lib().mi355_adam_step(C.cast(slots, C.c_void_p), len(ks), self.beta_1, self.beta_2, self.epsilon,
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        for k, n in zip(ks, new):
            self.variables[k] = n.requires_grad_(True)

    def reset(self, key):
        import torch
        z = torch.zeros_like(self.variables[key].detach())
        self.state[key] = (z, z.clone())
        self.t[key] = 0
