"""Data-parallel gradient exchange: bucketed all-reduce overlapped with backward.

Replaces DistributedDataParallel(find_unused_parameters=True) at reference main.py:172 (SURVEY.md
section 2.3, C1).  One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm and
runs its collectives on its own HIP stream, so a bucket's all-reduce overlaps the backward kernels
still being enqueued on the compute stream.  Design for xGMI (7 point-to-point links per GPU):
few large buckets (default 64 MB) in reverse-registration order; gradients live INSIDE the flat
bucket buffers (param.grad is a view), so there is no pack/unpack copy and the collective moves
exactly the gradient bytes once.

Parameters that never receive a gradient (e.g. the unused `backbone.0.body.head.*`, the reason the
reference needs find_unused_parameters=True) cannot hang anything: buckets that did not fill during
backward are reduced by `finish()`.
"""
import os

import torch
import torch.distributed as dist


class _EventWork:
    """`work.wait()` of a collective issued through spe_amd.comm: the current stream waits for the collective's event."""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class GradAllReducer:
    def __init__(self, params, bucket_bytes=64 << 20, average=True, group=None, flatten_params=False, broadcast=True,
                 always_reduce=False, wire_dtype=None, buffers=(), comm=None, cu_reserve=None):
        """flatten_params: also move the parameters themselves into flat per-bucket buffers with the gradient layout
        (param.data becomes a view) - what spe_amd.optim.FlatAdamW steps in one launch per bucket.  Construct the
        reducer AFTER the model is on its device: `module.to(...)` / `.cuda()` re-allocates parameters and would
        detach them from the flat buffers (load_state_dict and in-place updates are fine).
        broadcast: rank 0's parameters (and `buffers`) overwrite every other rank's at construction, as
        DistributedDataParallel does (reference main.py:161-172 seeds each rank with seed + rank BEFORE build_model, so
        everything that is not loaded from a checkpoint starts different per rank).
        always_reduce: issue the collectives even in a one-rank group (tests of the RCCL path on one GPU).
        wire_dtype: torch.bfloat16 sends the buckets in bf16 (half the xGMI bytes, one conversion pass each way); the
        default None keeps DDP's fp32 gradients.
        comm: a spe_amd.comm.RcclComm - the collectives then go through the C ABI of libspe_comm.so (include/spe_comm.h)
        on its side stream instead of torch.distributed (same RCCL underneath).
        cu_reserve: workgroup slots left free for the RCCL ring that runs beside the backward (kernels.set_cu_reserve: the
        single-round attention grids shrink from 512 to 512 - cu_reserve workgroups).  Default for world > 1: the channel cap
        NCCL_MAX_NCHANNELS (bench.py sets 32 before the process group is created), else 32; 0 in a one-rank job.  Measured with
        tools/dp_proxy.py (a one-GPU proxy, not RCCL): foreign workgroups beside the step cost +9 % with the solo grids and
        +3 % with 32 slots reserved (profiles/r03_dp_proxy.json)."""
        self.params = [p for p in params if p.requires_grad]
        self.flatten_params = flatten_params
        self.group = group
        self.comm = comm
        self.initialised = comm is not None or (dist.is_available() and dist.is_initialized())
        self.world = comm.world if comm is not None else (dist.get_world_size(group) if self.initialised else 1)
        self.collective = self.world > 1 or (always_reduce and self.initialised)
        if cu_reserve is None:
            cu_reserve = int(os.environ.get("NCCL_MAX_NCHANNELS", "32")) if self.world > 1 else 0
        self.cu_reserve = cu_reserve
        self._cu_reserve_prev = None
        if self.params and self.params[0].is_cuda:
            from . import kernels as _K
            self._cu_reserve_prev = _K.get_cu_reserve()        # restored by remove(): the grids are global to the process
            _K.set_cu_reserve(cu_reserve)
        self.average = average
        # set by FlatAdamW (and only by it): the 1/world is folded into its update launch (no div_ per bucket); p.grad / the
        # buckets then hold the world SUM after finish() - see grad_scale() / averaged_grad()
        self.average_in_optimizer = False
        self.wire_dtype = wire_dtype
        self.measure = False                   # bench: record events around the waits of finish() (exposed all-reduce time)
        self.exposed_ms = []
        self._wait_events = []
        self.buckets = []          # dicts: flat, params, pending, work
        self._bucket_of = {}
        self._views = {}
        cur, cur_bytes = [], 0
        for p in reversed(self.params):                     # ~ order in which autograd finishes them
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._make_bucket(cur)
        if broadcast and self.world > 1:
            with torch.no_grad():
                bc = (lambda t: comm.broadcast(t, 0)) if comm is not None else (lambda t: dist.broadcast(t, src=0, group=group))
                if flatten_params:
                    for b in self.buckets:
                        bc(b["flat_p"])
                else:
                    for p in self.params:
                        bc(p.data)
                for t in buffers:
                    bc(t)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        # bias / LayerNorm / LayerScale gradients are sums over workgroups: with the buckets registered their kernels only leave partial
        # rows and ONE launch adds them for many producers (kernels.defer_reductions); flushed before a bucket goes out, in finish() and
        # - by kernels.backward_scope - at the end of every backward pass
        # A parameter used by SEVERAL nodes of one graph (the decoder's final LayerNorm, applied to every layer's output) must not
        # take part: autograd adds the nodes' gradients itself and would read the bucket view before the deferred part has landed.
        # Such parameters are learned in the first step (deferral still off; static graph, like the unused ones): they never get a
        # bucket view again (_spe_shared), and the buckets are registered at the end of the first finish().
        self._defer = False
        self._defer_wanted = bool(self.params) and self.params[0].is_cuda and len(self.buckets) <= 64
        # Parameters that received no gradient in the first step (e.g. `backbone.0.body.head.*`) are treated as
        # statically unused afterwards (cf. DDP static_graph): their bucket no longer waits for them, so it - and,
        # because collectives go out strictly in bucket order, every later bucket - can start during backward
        # instead of in finish().  Every rank runs the same model, so every rank learns the same set.
        self.learn_unused = True
        self._fired = set()
        self._static_unused = None
        self._zero_views = None
        self.reset()

    def _make_bucket(self, plist):
        dev, dt = plist[0].device, plist[0].dtype
        al = lambda k: (k + 63) & ~63                        # 256-B aligned views: vectorised kernels write into them
        n = sum(al(p.numel()) for p in plist)
        flat = torch.zeros(n, device=dev, dtype=dt)
        flat_p = torch.zeros(n, device=dev, dtype=dt) if self.flatten_params else None
        off = 0
        offsets = []
        for p in plist:
            v = flat[off:off + p.numel()].view_as(p)
            self._views[p] = v
            p._spe_grad_buf = v                              # spe_amd backward kernels write here (kernels.grad_buffer)
            if flat_p is not None:
                pv = flat_p[off:off + p.numel()].view_as(p)
                with torch.no_grad():
                    pv.copy_(p.data)
                p.data = pv
            offsets.append((p, off))
            off += al(p.numel())
            self._bucket_of[p] = len(self.buckets)
        # wire_dtype: the send / receive buffer of the bucket in the wire format, allocated once (not per step)
        wire = torch.empty(n, device=dev, dtype=self.wire_dtype) if self.wire_dtype is not None else None
        self.buckets.append({"flat": flat, "flat_p": flat_p, "offsets": offsets, "params": list(plist), "pending": 0,
                             "work": None, "wire": wire})

    # Gradients of parameters with more than ZERO_MAX elements are always OVERWRITTEN by their producer (GEMM stores, split-K slab
    # sums with accumulate = 0, or the copy in _on_grad); only the small ones (biases, LayerNorm / LayerScale vectors, head mixers:
    # their kernels add partial sums atomically into a zeroed view) and the statically unused ones need zeros.
    ZERO_MAX = 16384         # == spe_amd.kernels.ACC_ZERO_MAX (kernels._zeros_or zeroes larger accumulate-into views itself)

    def reset(self):
        """Call before every backward (after the optimizer consumed the gradients): re-arm the buckets and detach the .grad
        views.  With .grad = None autograd adopts the first incoming gradient tensor instead of adding it: kernels that wrote
        into the bucket view (kernels.grad_buffer) cost nothing extra, any other gradient is moved into the bucket by
        `_on_grad`; a parameter that gets no gradient ends the step with zeros in the bucket.
        First step: the whole buckets are zeroed (330 MB at cfg2).  From the second step on only the views that need it are:
        small accumulate-into gradients and statically unused parameters (< 1 % of the bytes, one multi-tensor launch); a large
        parameter that unexpectedly received no gradient is zeroed in finish()."""
        self._flush_deferred()      # sums a failed backward left pending land in the OLD gradients, not on top of the new ones
        skip = self._static_unused or ()
        if self._static_unused is None:
            for b in self.buckets:
                b["flat"].zero_()
        else:
            if self._zero_views is None:
                self._zero_views = [self._views[p] for p in self.params if p.numel() <= self.ZERO_MAX or p in skip]
            if self._zero_views:
                torch._foreach_zero_(self._zero_views)
        for b in self.buckets:
            b["pending"] = sum(1 for p in b["params"] if p not in skip)
            b["work"] = None
            for p in b["params"]:
                p.grad = None
                p._spe_grad_fresh = True
                p._spe_handed = False
        self._next = 0
        self._fired = set()
        if self._defer:
            from . import kernels as _K
            _K._FLUSH_QUEUED[0] = False        # (a backward that raised may have left its end-of-pass flush queued and never run)

    def _flush_deferred(self):
        if self._defer:
            from . import kernels as _K
            _K.reduce_flush()

    def zero_grad(self):
        self.reset()

    def _launch(self, b):
        self._flush_deferred()                 # the bucket's deferred sums land before it is reduced / handed to the optimiser
        if self.collective:
            buf = b["flat"]
            if self.wire_dtype is not None:
                buf = b["wire"]
                buf.copy_(b["flat"])
            if self.comm is not None:
                b["work"] = _EventWork(self.comm.all_reduce_async(buf))
            else:
                b["work"] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b["work"] = "local"

    def _on_grad(self, p):
        b = self.buckets[self._bucket_of[p]]
        view = self._views[p]
        if self._static_unused and p in self._static_unused:
            if b["work"] is not None:
                raise RuntimeError("GradAllReducer: a parameter that got no gradient in the first step received one "
                                   "after its bucket was reduced; construct the reducer with learn_unused = False")
            self._static_unused = self._static_unused - {p}      # used after all: count it from the next step on
            self._zero_views = None
            b["pending"] += 1
        if p.grad.data_ptr() != view.data_ptr():
            # the gradient was produced outside the bucket (torch op, or a parameter used twice whose contributions
            # autograd summed into a temporary): AccumulateGrad runs once per backward with the TOTAL, so copy it in
            if getattr(p, "_spe_handed", False) and not getattr(p, "_spe_shared", False):
                # its bucket view HAD been handed to a kernel: the parameter is used by more than one node
                p._spe_shared = True
                if self._defer:
                    import warnings
                    warnings.warn("GradAllReducer: a parameter turned out to be shared between graph nodes after the first step; "
                                  "its gradient of this step may miss a deferred bias / LayerNorm sum (construct the reducer "
                                  "before the first backward of the final graph, or set SPE_DEFER_REDUCE=0)")
            view.copy_(p.grad)
            p._spe_grad_fresh = False
            p.grad = view
        self._fired.add(p)
        b["pending"] -= 1
        if b["pending"] < 0:
            raise RuntimeError("GradAllReducer: a gradient arrived for a bucket that was not re-armed - call reset() (or the "
                               "optimizer's zero_grad()) before every backward")
        # collectives are issued strictly in bucket order so every rank enqueues the same sequence
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def finish(self):
        """Reduce buckets that never filled (unused parameters), wait for every collective, average."""
        self._flush_deferred()
        if self._static_unused is not None:
            # a LARGE parameter that fired in the first step but not in this one still holds the previous step's gradient (its
            # view is not zeroed by reset()): zero it before its bucket goes out.  Such a bucket cannot have been launched yet -
            # it was still waiting for this parameter.
            for b in self.buckets[self._next:]:
                for p in b["params"]:
                    if p not in self._fired and p not in self._static_unused and p.numel() > self.ZERO_MAX:
                        self._views[p].zero_()
        while self._next < len(self.buckets):
            self._launch(self.buckets[self._next])
            self._next += 1
        if self.learn_unused and self._static_unused is None:
            self._static_unused = frozenset(p for p in self.params if p not in self._fired)
        if self._defer_wanted and not self._defer:          # end of the first step: shared parameters are known now
            from . import kernels as _K
            _K.defer_reductions([b["flat"] for b in self.buckets])
            self._defer = _K._DEFER_FLATS is not None
            self._defer_wanted = False
        ev = None
        if self.measure and self.collective:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for b in self.buckets:
            if b["work"] not in (None, "local"):
                b["work"].wait()
                if self.wire_dtype is not None:
                    b["flat"].copy_(b["wire"])
            for p in b["params"]:
                if p.grad is None:                            # unused this step: zeros, like DDP's unused-parameter path
                    p.grad = self._views[p]
        if ev is not None:          # compute-stream time between "backward enqueued" and "last collective done"
            ev[1].record()
            self._wait_events.append(ev)
        if self.average and self.world > 1 and not self.average_in_optimizer:
            torch._foreach_div_([b["flat"] for b in self.buckets], float(self.world))

    def averaged_grad(self, p):
        """The gradient of `p` as DDP would leave it in p.grad (the mean over ranks), whatever the averaging mode: with a
        FlatAdamW attached the buckets hold the world SUM after finish() and the 1/world is applied inside its update launch."""
        return self._views[p] * self.grad_scale()

    def grad_scale(self):
        """Factor the optimizer still has to apply to the bucket contents (1/world when the averaging is folded in)."""
        return 1.0 / self.world if (self.average and self.average_in_optimizer and self.world > 1) else 1.0

    def exposed_ms_mean(self):
        """Mean compute-stream stall of finish() in ms (call after a device synchronisation; needs measure = True)."""
        ms = [a.elapsed_time(b) for a, b in self._wait_events]
        return sum(ms) / len(ms) if ms else 0.0

    def remove(self):
        for h in self._hooks:
            h.remove()
        for p in self.params:           # per-parameter marks of THIS reducer do not survive into the next one
            for a in ("_spe_shared", "_spe_handed", "_spe_grad_fresh", "_spe_grad_buf"):
                if hasattr(p, a):
                    delattr(p, a)
        if self._defer:
            from . import kernels as _K
            if _K._DEFER_FLATS is not None and _K._DEFER_FLATS and _K._DEFER_FLATS[0] is self.buckets[0]["flat"]:
                _K.defer_reductions(None)
            self._defer = False
        if self._cu_reserve_prev is not None:
            from . import kernels as _K
            _K.set_cu_reserve(self._cu_reserve_prev)
            self._cu_reserve_prev = None
