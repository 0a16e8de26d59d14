"""Optimiser step of the SPE training loop (reference engine.py:161-165 + main.py:177-191):

    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)      # 0.1 in the reference scripts
    optimizer.step()                                                  # torch.optim.AdamW, 3 LR groups

as two HIP launches per gradient bucket (csrc/optim.hip) on the flat buffers of spe_amd.dp.GradAllReducer
(flatten_params=True): parameters, gradients and both Adam moments share one layout, the global gradient norm is
reduced from per-block partial sums inside the update kernel (no host round trip), and the per-group learning
rate / weight decay come from a small segment table.  FlatAdamW is a torch.optim.Optimizer: `param_groups`,
LR schedulers (StepLR in the reference) and state_dict()/load_state_dict() keep working; the per-parameter state
entries are views into the flat moment buffers.
"""
import torch

from . import kernels as K
from .util.misc import host_to_device

_NORM_BLOCKS = 256


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None,
                 write_clipped_grads=True):
        if not getattr(reducer, "flatten_params", False):
            raise ValueError("FlatAdamW needs GradAllReducer(..., flatten_params=True)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.reducer = reducer
        # Gradient averaging (1/world) rides on the update launch.  This changes what the reducer's buckets (and p.grad) hold
        # after finish(): the world SUM, not DDP's mean - any other consumer must multiply by reducer.grad_scale() (or read
        # reducer.averaged_grad(p)).  A reducer can therefore serve ONE FlatAdamW and nothing that expects means.
        if getattr(reducer, "_folded_into", None) not in (None, id(self)):
            raise ValueError("FlatAdamW: this GradAllReducer already folds its averaging into another optimizer")
        reducer.average_in_optimizer = True
        reducer._folded_into = id(self)
        self.max_grad_norm = max_grad_norm
        self.write_clipped_grads = write_clipped_grads
        gid = {}
        for gi, grp in enumerate(self.param_groups):
            if tuple(grp["betas"]) != tuple(self.param_groups[0]["betas"]) or grp["eps"] != self.param_groups[0]["eps"]:
                raise ValueError("FlatAdamW: betas / eps must be the same for all parameter groups")
            for p in grp["params"]:
                gid[p] = gi
        self._step = 0
        # ONE 0-dim step tensor shared by every parameter's state entry (torch.optim.AdamW semantics: optimizer.state[p]["step"] is always
        # current) and bumped by a single host op per step() - not 717 tensor increments
        self._step_t = torch.zeros((), dtype=torch.float32)
        self._buckets = []
        for b in reducer.buckets:
            n = b["flat"].numel()
            m, v = torch.zeros_like(b["flat"]), torch.zeros_like(b["flat"])
            # runs of consecutive parameters of one group -> (segment end, group); padding follows its parameter
            ends, groups = [], []
            offs = b["offsets"]
            for i, (p, off) in enumerate(offs):
                if p not in gid:
                    raise ValueError("FlatAdamW: a parameter of the reducer is in no parameter group")
                end = offs[i + 1][1] if i + 1 < len(offs) else n
                if groups and groups[-1] == gid[p]:
                    ends[-1] = end
                else:
                    ends.append(end); groups.append(gid[p])
                st = self.state[p]
                st["step"] = self._step_t
                st["exp_avg"] = m[off:off + p.numel()].view_as(p)
                st["exp_avg_sq"] = v[off:off + p.numel()].view_as(p)
            if len(ends) > 64:
                raise ValueError("FlatAdamW: more than 64 parameter-group runs in one bucket")
            self._buckets.append({"b": b, "m": m, "v": v, "groups": groups,
                                  "seg_end": host_to_device(ends, torch.int64, m.device), "tab": None, "tab_key": None})
        dev = reducer.buckets[0]["flat"].device
        self._partials = torch.zeros((_NORM_BLOCKS * len(self._buckets),), device=dev, dtype=torch.float32)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._step += 1
        self._step_t += 1
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        bc1, bc2 = 1.0 - b1 ** self._step, 1.0 - b2 ** self._step
        clip = float(self.max_grad_norm) if self.max_grad_norm else 0.0
        if clip > 0:
            for i, e in enumerate(self._buckets):
                K.sqnorm_partials(e["b"]["flat"], self._partials[i * _NORM_BLOCKS:(i + 1) * _NORM_BLOCKS])
        for e in self._buckets:
            key = tuple((self.param_groups[g]["lr"], self.param_groups[g]["weight_decay"]) for g in e["groups"])
            if key != e["tab_key"]:            # LR scheduler moved: refresh the (tiny) per-segment table
                dev = e["m"].device
                e["tab"] = (host_to_device([k[0] for k in key], torch.float32, dev),
                            host_to_device([k[1] for k in key], torch.float32, dev))
                e["tab_key"] = key
            K.adamw_flat(e["b"]["flat_p"], e["b"]["flat"], e["m"], e["v"], e["seg_end"], e["tab"][0], e["tab"][1],
                         b1, b2, eps, bc1, bc2, self._partials, clip, self.write_clipped_grads,
                         grad_scale=self.reducer.grad_scale())
        K.weights_changed()            # the update bypassed autograd's version counters: drop cached bf16 weight copies
        return loss

    def state_dict(self):
        """torch's format.  Inside this optimizer every parameter's `step` entry is the one shared, always current tensor; the EXPORTED
        dict gives each parameter its own copy: pickle / torch.save keep aliasing, and torch.optim.AdamW (the reference's optimizer,
        main.py:189-191 and its resume path :224-232) bumps every entry of the list it is handed - one shared tensor would advance by
        the parameter count per step after such a resume."""
        sd = super().state_dict()
        for st in sd["state"].values():
            if "step" in st:
                st["step"] = self._step_t.clone()
        return sd

    def zero_grad(self, set_to_none=True):
        """The reference loop calls optimizer.zero_grad() before backward (engine.py:161): the gradients live in the
        reducer's buckets, so this re-arms them (zeroes the buckets, detaches .grad, resets the bucket counters)."""
        self.reducer.reset()

    def load_state_dict(self, state_dict):
        """Standard torch format; the moments are copied INTO the flat buffers (the views must stay views)."""
        views = {p: (self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for g in self.param_groups for p in g["params"]}
        super().load_state_dict(state_dict)
        steps = []
        for g in self.param_groups:
            for p in g["params"]:
                st = self.state[p]
                m, v = views[p]
                if "exp_avg" in st and st["exp_avg"].data_ptr() != m.data_ptr():
                    m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = m, v
                if "step" in st:
                    steps.append(int(st["step"]))
        if steps:
            self._step = max(steps)
        self._step_t.fill_(float(self._step))
        for g in self.param_groups:                 # re-share the step tensor (load_state_dict gave every parameter its own copy)
            for p in g["params"]:
                self.state[p]["step"] = self._step_t
