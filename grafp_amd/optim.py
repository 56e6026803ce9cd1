"""The optimizer update of the training step on the hand-written multi-tensor kernel (csrc/adam.hip).

`Adam` IS a torch.optim.Adam (same constructor, param_groups, state keys `step` / `exp_avg` / `exp_avg_sq`, state_dict
format -- a checkpoint written by /root/reference/train.py:212-220 loads, and one written here loads into torch's class),
with `step()` replaced for what /root/reference/train.py:174 uses: no weight decay, no amsgrad, no maximize.  Anything else
raises -- there is no second implementation behind it.
"""
import ctypes

import torch

from ._lib import check, lib


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, **kw):
        for k in ("weight_decay", "amsgrad", "maximize", "differentiable"):
            if kw.get(k):
                raise NotImplementedError(f"grafp_amd.optim.Adam: {k} is outside the path (train.py:174 uses the defaults)")
        kw.pop("fused", None)
        kw.pop("foreach", None)
        # capturable = the step counters live on the device (torch's own convention for this flag): state_dict compatible
        super().__init__(params, lr=lr, betas=betas, eps=eps, capturable=True, foreach=False, **kw)
        self._tables = None         # per group: ctypes arrays of the pointers that do not change between steps

    # -- state ----------------------------------------------------------------------------------------------------------
    def _group_state(self, gi, group):
        """Lazily creates the state of the group's parameters exactly as torch's Adam does (step: f32 scalar on the
        device; exp_avg / exp_avg_sq: zeros like the parameter) -- with ONE difference that no reader can see: the step
        counters of a group are 0-d views of one flat tensor, so that bumping all of them is one launch."""
        ps = [p for p in group["params"] if p.requires_grad]
        if not ps:
            return None                                          # a group of frozen parameters: nothing to update
        cached = self._tables[gi] if self._tables is not None and gi < len(self._tables) else None
        if cached is not None and cached["n"] == len(ps) and cached["p"][0] == ps[0].data_ptr() and all(
                self.state[p].get("step") is not None and self.state[p]["step"].data_ptr() == cached["steps"].data_ptr() + 4 * i
                for i, p in ((0, ps[0]), (len(ps) - 1, ps[-1]))):
            return cached
        dev = ps[0].device
        for p in ps:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise RuntimeError("grafp_amd.optim.Adam: contiguous f32 parameters on a HIP device only (no CPU fallback)")
        steps = torch.zeros(len(ps), dtype=torch.float32, device=dev)
        for i, p in enumerate(ps):
            st = self.state[p]
            if "step" in st:                      # a loaded checkpoint (or torch's own state): keep the values
                steps[i] = float(st["step"]) if not torch.is_tensor(st["step"]) else st["step"].to(dev, torch.float32)
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            for k in ("exp_avg", "exp_avg_sq"):
                if st[k].dtype != torch.float32 or st[k].device != dev or not st[k].is_contiguous():
                    st[k] = st[k].to(dev, torch.float32).contiguous()
            st["step"] = steps[i]
        n = len(ps)
        vp = ctypes.c_void_p
        tab = {"n": n, "ps": ps, "steps": steps,
               "p": (vp * n)(*[p.data_ptr() for p in ps]),
               "m": (vp * n)(*[self.state[p]["exp_avg"].data_ptr() for p in ps]),
               "v": (vp * n)(*[self.state[p]["exp_avg_sq"].data_ptr() for p in ps]),
               "s": (vp * n)(*[steps.data_ptr() + 4 * i for i in range(n)]),
               "numel": (ctypes.c_int64 * n)(*[p.numel() for p in ps]),
               "g": (vp * n)(), "sub": {}}
        if self._tables is None:
            self._tables = []
        while len(self._tables) <= gi:
            self._tables.append(None)
        self._tables[gi] = tab
        return tab

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        stream = ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        vp = ctypes.c_void_p
        for gi, group in enumerate(self.param_groups):
            if group.get("weight_decay") or group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("grafp_amd.optim.Adam: weight_decay / amsgrad / maximize are outside the path")
            tab = self._group_state(gi, group)
            if tab is None:
                continue
            ps = tab["ps"]
            grads = [p.grad for p in ps]
            if all(g is not None for g in grads):
                sub, n = tab, tab["n"]
                tab["steps"].add_(1.0)
            else:
                # a parameter that took no part in the step is skipped, as torch skips it (its counter does not advance):
                # tables over the participating subset, cached by which parameters those are
                idx = tuple(i for i, g in enumerate(grads) if g is not None)
                if not idx:
                    continue
                sub = tab["sub"].get(idx)
                if sub is None:
                    n = len(idx)
                    sub = {"p": (vp * n)(*[tab["p"][i] for i in idx]), "m": (vp * n)(*[tab["m"][i] for i in idx]),
                           "v": (vp * n)(*[tab["v"][i] for i in idx]), "s": (vp * n)(*[tab["s"][i] for i in idx]),
                           "numel": (ctypes.c_int64 * n)(*[tab["numel"][i] for i in idx]), "g": (vp * n)(),
                           "sel": torch.tensor(idx, dtype=torch.int64, device=tab["steps"].device),
                           "one": torch.ones(n, dtype=torch.float32, device=tab["steps"].device)}
                    tab["sub"] = {idx: sub}                      # (one cached subset: the usual case is a fixed one)
                n = len(idx)
                grads = [grads[i] for i in idx]
                tab["steps"].index_add_(0, sub["sel"], sub["one"])
            g_arr = sub["g"]
            for i, g in enumerate(grads):
                if g.dtype != torch.float32 or not g.is_contiguous():
                    raise RuntimeError("grafp_amd.optim.Adam: contiguous f32 gradients only")
                g_arr[i] = g.data_ptr()
            lr = group["lr"]
            lr_dev = vp(lr.data_ptr()) if torch.is_tensor(lr) and lr.is_cuda else None
            beta1, beta2 = group["betas"]
            check(lib.grafp_adam_multi_f32(sub["p"], g_arr, sub["m"], sub["v"], sub["s"], sub["numel"], n, lr_dev,
                                           0.0 if lr_dev is not None else float(lr), float(beta1), float(beta2),
                                           float(group["eps"]), stream), "adam_multi")
        return loss
