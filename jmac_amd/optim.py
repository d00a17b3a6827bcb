"""Adam for the training step (train.py:406-407: ``torch.optim.Adam(model.parameters(), lr=...)``; :358-359 one ``step()`` per batch)
as ONE launch over all parameter tensors: ``jmac_adam_step_f32`` (csrc/optim.hip).

Drop-in for ``torch.optim.Adam`` / ``torch.optim.AdamW`` on fp32 HIP parameters with dense gradients: same constructor arguments and
defaults, same update, same ``state_dict`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, so a checkpoint written by
either loads into the other).  The step count is a device tensor, so a step captured in a hipGraph replays correctly (torch needs
``capturable=True`` for that); ``amsgrad`` is not offered.  ``lr``, the betas, ``eps`` and ``weight_decay`` are HOST numbers passed as
kernel arguments: a captured step replays the values it was captured with, so a learning-rate schedule needs a re-capture after
every change (torch's capturable Adam takes a tensor ``lr`` instead; the reference never schedules: train.py:406-407).  Why it exists: torch's fused multi-tensor kernel gives each 65 536-element
chunk to one workgroup, 125 workgroups for the 6.3 M parameters of the DBP-5L model -- under half of the MI355X's 256 CUs.
"""
from __future__ import annotations

import torch

from ._lib import AdamTask, check, lib, ptr, stream


class Adam(torch.optim.Optimizer):
    _decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *, maximize=False):
        if amsgrad:
            raise NotImplementedError("jmac_amd.optim.Adam: amsgrad is not offered (the reference trains without it: train.py:406-407)")
        if isinstance(lr, torch.Tensor):
            raise TypeError("jmac_amd.optim.Adam: lr is a host number (it is a kernel argument)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= weight_decay or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("jmac_amd.optim.Adam: invalid hyper-parameter (lr=%r betas=%r eps=%r weight_decay=%r)"
                             % (lr, betas, eps, weight_decay))
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize))
        self._managed = {}        # id(step tensor) -> [step tensor, ids of the parameters that share it, aux, betas of aux]
        self._tables = {}         # id(step tensor) -> (key, ctypes task table)
        self._keep = []

    # torch keeps one step count PER PARAMETER: a parameter that gets no gradient in a call is skipped and its count stands still
    # (train.py steps two optimizers over model.parameters(), each loss reaching its own subset).  Here parameters that have always
    # stepped together share one device count -- one launch covers them -- and a subset that steps alone is split off with a copy.
    @staticmethod
    def _aux(v, betas, dev):
        """{beta1^v, beta2^v, 0}: the running powers of jmac_adam_step_f32 for a count v, and its zero-at-rest word."""
        return torch.tensor([betas[0] ** v, betas[1] ** v, 0.0], dtype=torch.float64).to(dev)

    def _counters(self, ps, dev, betas):
        capturing = torch.cuda.is_current_stream_capturing()
        adopt = {}
        for p in ps:
            st = self.state[p]
            c = st.get("step")
            if c is not None and id(c) in self._managed and self._managed[id(c)][0] is c:
                continue
            if capturing:
                raise RuntimeError("jmac_amd.optim.Adam: the first step of a parameter (or the first after load_state_dict) reads "
                                   "its step count on the host: run one step outside the hipGraph capture first")
            v = 0.0 if c is None else float(c)              # fresh state, or a count loaded from a torch.optim.Adam checkpoint
            cn = adopt.get(v)
            if cn is None:
                cn = adopt[v] = torch.full((), v, dtype=torch.float32, device=dev)
                self._managed[id(cn)] = [cn, set(), self._aux(v, betas, dev), betas]
            self._managed[id(cn)][1].add(id(p))
            st["step"] = cn
        groups = {}
        for p in ps:
            c = self.state[p]["step"]
            groups.setdefault(id(c), (c, []))[1].append(p)
        out = []
        for cid, (c, plist) in groups.items():
            members = self._managed[cid][1]
            if len(plist) != len(members):                  # only some of the sharers step now: they continue on their own count
                if capturing:
                    raise RuntimeError("jmac_amd.optim.Adam: the set of parameters with gradients changed inside a hipGraph capture")
                c2 = c.clone()
                mine = set(id(p) for p in plist)
                self._managed[id(c2)] = [c2, mine, self._managed[cid][2].clone(), self._managed[cid][3]]
                members -= mine
                for p in plist:
                    self.state[p]["step"] = c2
                c = c2
            rec = self._managed[id(c)]
            if rec[3] != betas:                             # the group's betas were changed: the powers restart from the count
                if capturing:
                    raise RuntimeError("jmac_amd.optim.Adam: betas changed inside a hipGraph capture")
                rec[2], rec[3] = self._aux(float(c), betas, dev), betas
            out.append((c, plist, rec[2]))
        return out

    def _group_step(self, group):
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return
        dev = ps[0].device
        for p in ps:
            if p.device != dev or not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                raise TypeError("jmac_amd.optim.Adam: fp32 parameters with dense fp32 gradients on one HIP device (there is no CPU path)")
            if not p.is_contiguous():
                raise ValueError("jmac_amd.optim.Adam: parameters must be contiguous")
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            for k in ("exp_avg", "exp_avg_sq"):
                if st[k].dtype != torch.float32 or not st[k].is_contiguous() or st[k].device != dev:
                    st[k] = st[k].to(device=dev, dtype=torch.float32).contiguous()
        b1, b2 = (float(b) for b in group["betas"])
        for counter, plist, aux in self._counters(ps, dev, (b1, b2)):
            tasks = []
            for p in plist:
                st = self.state[p]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                tasks.append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            key = tuple((t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[0].numel()) for t in tasks)
            cached = self._tables.get(id(counter))
            if cached is None or cached[0] != key:
                table = (AdamTask * len(tasks))()
                for i, (p, g, m, v) in enumerate(tasks):
                    table[i].p, table[i].g, table[i].m, table[i].v, table[i].n, table[i].vec4 = ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), 0
                cached = self._tables[id(counter)] = (key, table)
            check(lib().jmac_adam_step_f32(cached[1], len(tasks), ptr(counter), ptr(aux), float(group["lr"]), b1, b2,
                                           float(group["eps"]), float(group["weight_decay"]), 1 if self._decoupled else 0,
                                           1 if group["maximize"] else 0, stream()), "jmac_adam_step_f32")
            self._keep.append(tasks)      # copies .contiguous() made of strided gradients stay alive until the next step

    def state_dict(self):
        """torch.optim.Adam's layout.  ``step`` leaves as one host tensor per parameter (torch's own default form): the shared device
        count of this class must not reach an optimizer that increments every parameter's ``step`` in place."""
        sd = super().state_dict()
        host = {}                                               # one blocking copy per DISTINCT counter (the parameters of a group share one)

        def to_host(t):
            key = (t.data_ptr(), t.device)
            if key not in host:
                host[key] = t.detach().to("cpu", copy=True)
            return host[key].clone()
        sd["state"] = {k: {**v, "step": to_host(v["step"])} if "step" in v else dict(v) for k, v in sd["state"].items()}
        return sd

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._keep = []
        for group in self.param_groups:
            self._group_step(group)
        return loss


class AdamW(Adam):
    """torch.optim.AdamW's decoupled weight decay (default 1e-2) on the same kernel."""
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize)
