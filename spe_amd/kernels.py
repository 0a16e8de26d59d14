"""Thin tensor-level wrappers over the C ABI (include/spe_hip.h).  No autograd here - see ops.py.

Every function takes CUDA fp32 tensors, allocates its outputs with torch (device memory is
PyTorch's job on this path) and enqueues the kernel on torch's current HIP stream.
"""
import ctypes
import os

import torch

from . import lib

# Precision modes (operand format of the MFMA products; accumulation, residual stream, LayerNorm / softmax statistics and the
# losses are fp32 in every mode):
#   0 "bf16"   single bf16 operands everywhere
#   1 "bf16x3" 3-term bf16 split (hi*hi + lo*hi + hi*lo, ~fp32) everywhere, fp32 materialised attention (csrc/talking.hip)
#   2 "bf16s"  FORWARD products on split operands (what north_star's 1e-3 on logits / losses needs), BACKWARD products on
#              single bf16 operands (gradients carry bf16 rounding like any mixed-precision trainer) - the benchmark mode
_PRECISION = 2       # default: the benchmark mode
import functools as _functools
import threading as _threading

_TLS = _threading.local()      # .in_bwd: set by @backward_scope around every autograd backward of spe_amd.ops - per THREAD (autograd
#                                runs backward on its own worker threads)


def _in_bwd():
    return getattr(_TLS, "in_bwd", False)


def backward_scope(fn):
    """Decorator of the `backward` staticmethods: products issued inside run at the mode's BACKWARD operand precision."""
    @_functools.wraps(fn)
    def wrapped(ctx, *grads):
        prev, _TLS.in_bwd = _in_bwd(), True
        if _DEFER_FLATS is not None and not _FLUSH_QUEUED[0]:
            # deferred sums (defer_reductions) of this backward pass land when the engine is done with it - before backward() /
            # autograd.grad() return to whoever reads the gradients
            _FLUSH_QUEUED[0] = True
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_flush_after_backward)
            except RuntimeError:               # not inside an engine run (a backward called by hand)
                _FLUSH_QUEUED[0] = False
        try:
            return fn(ctx, *grads)
        finally:
            _TLS.in_bwd = prev
    return wrapped


_FLUSH_QUEUED = [False]
_DEFER_FLATS = None        # strong references: a registered address range must not be freed and handed to another tensor


def _flush_after_backward():
    _FLUSH_QUEUED[0] = False
    reduce_flush()


def forward_scope(fn):
    """Decorator of the `forward` staticmethods: a forward entered from INSIDE a backward (torch.utils.checkpoint recomputes the
    block in the saved-tensor unpack hook of the wrapped backward) runs at the mode's FORWARD operand precision, so the recomputed
    activations equal the original ones."""
    @_functools.wraps(fn)
    def wrapped(ctx, *args):
        prev, _TLS.in_bwd = _in_bwd(), False
        try:
            return fn(ctx, *args)
        finally:
            _TLS.in_bwd = prev
    return wrapped


def split_now():
    """True when the product about to be issued takes split (hi + lo) bf16 operands."""
    return _PRECISION == 1 or (_PRECISION == 2 and not _in_bwd())


def split_fwd():
    """bf16s forward: the bf16-copy GEMMs run on (hi, lo) operand pairs and every producer of such an operand also writes lo."""
    return _PRECISION == 2 and not _in_bwd()
# Philox stream for dropout: (seed, running offset).  Each dropout site draws a fresh offset.
# seed None = not set by the caller: derived on first use from torch's seed (the reference's main.py:161-164 seeds torch
# with args.seed + rank and nothing else, so its dropout masks differ per rank and per run; ours then do too).
_RNG = {"seed": None, "offset": 0}


def set_precision(mode):
    global _PRECISION
    _PRECISION = {"bf16": 0, "bf16x3": 1, "bf16s": 2, 0: 0, 1: 1, 2: 2}[mode]


def get_precision():
    return ("bf16", "bf16x3", "bf16s")[_PRECISION]


def manual_seed(seed):
    _RNG["seed"] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _RNG["offset"] = 0


def next_rng():
    """-> (seed, offset) for one dropout site; offsets never repeat within a process."""
    if _RNG["seed"] is None:
        _RNG["seed"] = (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0x5EEDC0DE) & 0xFFFFFFFFFFFFFFFF
    _RNG["offset"] += 1
    return _RNG["seed"], _RNG["offset"]


# Optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg).
_TIMED = {}


def enable_timing(names):
    """Record a (start, stop) event pair around every launch of the named entry points."""
    _TIMED.clear()
    for n in names:
        _TIMED[n] = []


def timing_results():
    """-> {name: (launches, mean_ms)}; call after torch.cuda.synchronize()."""
    return {n: (len(ev), (sum(a.elapsed_time(b) for a, b in ev) / len(ev)) if ev else 0.0) for n, ev in _TIMED.items()}


# entry point -> positions of (M, N, K) in its argument list: GEMM launches can be timed per problem shape, e.g.
# enable_timing(["spe_gemm_bf16nt:8300,384,384"])
_GEMM_DIMS = {"spe_gemm_bf16nt": (7, 8, 9), "spe_gemm_bf16nt_ex": (16, 17, 18)}


# Reduction workspace of the deterministic cross-workgroup sums (include/spe_hip.h: spe_set_reduce_workspace; csrc/det_reduce.h):
# 16 MiB of device memory registered with the library on the first launch of this process (one process per GPU).
_RWS = None
_RWS_BYTES = 16 << 20


_RWS_DEVIDX = -1
_RWS_STREAM = None     # raw handle of the stream the last launch went to: the ticket / slab workspace is only safe for stream-ordered launches


def _register_reduce_ws():
    global _RWS
    dev = torch.device("cuda", torch.cuda.current_device())
    _RWS = torch.zeros((_RWS_BYTES,), device=dev, dtype=torch.uint8)
    lib.call("spe_set_reduce_workspace", _RWS.data_ptr(), _RWS_BYTES, _st())


def _guard_reduce_ws(cur, st):
    """Slow path of _st(): the launch stream or device changed.  The deterministic reductions (csrc/det_reduce.h) share ONE ticket /
    slab workspace per process: launches must come from the device it lives on (one process per GPU) and be stream-ordered.  A
    launch from another stream first waits for everything the previous stream was given (an event edge, once per switch); another
    device is an error."""
    global _RWS_STREAM, _RWS_DEVIDX
    if _RWS is None:
        _register_reduce_ws()
    if cur != _RWS.device.index:
        raise lib.SpeLibraryError(f"spe_amd kernels were first used on cuda:{_RWS.device.index} and are now launched on cuda:{cur}: "
                                  "the reduction workspace is per process (one process per GPU)")
    if _RWS_STREAM is not None and st != _RWS_STREAM:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.ExternalStream(_RWS_STREAM, device=_RWS.device) if _RWS_STREAM else torch.cuda.default_stream(_RWS.device))
        torch.cuda.current_stream().wait_event(ev)
    _RWS_STREAM, _RWS_DEVIDX = st, cur


# ---- deferred cross-workgroup sums (include/spe_hip.h: spe_reduce_defer_*; csrc/det_reduce.h) ------------------------------------
# Sums whose destination is a parameter gradient inside the registered all-reduce buckets are left as per-workgroup partial rows and
# added by ONE flush launch for many producers (the fixed-order tree costs every producer 5-8 us).  The reducer registers its buckets
# and flushes before a bucket goes out / at the end of the backward.
DEFER_REDUCE = os.environ.get("SPE_DEFER_REDUCE", "1") != "0"      # developer knob (A/B)
_DEFER_ARENA = None


def defer_reductions(flats, arena_bytes=48 << 20):
    """flats: the flat gradient buffers (fp32 CUDA tensors) whose views may receive deferred sums; None / empty: deferral off."""
    global _DEFER_ARENA, _DEFER_FLATS
    if _DEFER_FLATS is not None:
        lib.call("spe_reduce_flush", _st())
    if not flats or not DEFER_REDUCE or not flats[0].is_cuda or flats[0].dtype != torch.float32:
        lib.call("spe_reduce_defer_ranges", None, None, 0)
        _DEFER_FLATS = None
        return
    if _DEFER_ARENA is None or _DEFER_ARENA.device != flats[0].device:
        _DEFER_ARENA = torch.empty((arena_bytes,), device=flats[0].device, dtype=torch.uint8)
        lib.call("spe_reduce_defer_arena", _DEFER_ARENA.data_ptr(), arena_bytes)
    n = len(flats)
    ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in flats])
    sizes = (ctypes.c_size_t * n)(*[f.numel() * 4 for f in flats])
    lib.call("spe_reduce_defer_ranges", ptrs, sizes, n)
    _DEFER_FLATS = list(flats)


def reduce_flush():
    """Add every pending deferred sum to its destination (one launch on the current stream; nothing pending: no launch)."""
    if _DEFER_FLATS is not None:
        lib.call("spe_reduce_flush", _st())


def _call(name, *args):
    if _RWS is None:
        _register_reduce_ws()
    if not _TIMED:
        return lib.call(name, *args)
    ev = _TIMED.get(name)
    if ev is None and name in _GEMM_DIMS:
        i, j, k = _GEMM_DIMS[name]
        ev = _TIMED.get("%s:%d,%d,%d" % (name, args[i], args[j], args[k]))
    if ev is None:
        return lib.call(name, *args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    lib.call(name, *args)
    b.record()
    ev.append((a, b))


# Small accumulate-into buffers (bias / gamma / weight-gradient sums) are carved out of a pre-zeroed arena: one
# memset per 16 MB instead of one fill launch per buffer (~600 launches per training step).  A carved view is
# handed out exactly once; when the arena is exhausted a fresh one is allocated (the old one lives as long as any
# view of it does).
_ZPOOL = {"buf": None, "off": 0}
_ZPOOL_FLOATS = 4 << 20


def zeros_small(n, device):
    if n > (1 << 18):
        return torch.zeros((n,), device=device, dtype=torch.float32)
    na = (n + 63) & ~63
    z = _ZPOOL
    if z["buf"] is None or z["buf"].device != device or z["off"] + na > z["buf"].numel():
        z["buf"] = torch.zeros((_ZPOOL_FLOATS,), device=device, dtype=torch.float32)
        z["off"] = 0
    out = z["buf"][z["off"]:z["off"] + n]
    z["off"] += na
    return out


def grad_buffer(param):
    """Zeroed, still unclaimed gradient storage of `param` inside its all-reduce bucket (spe_amd.dp.GradAllReducer
    publishes it as `param._spe_grad_buf` and re-arms `_spe_grad_fresh` every step), or None.  A backward kernel
    that writes its parameter gradient straight into this view - and returns it - makes autograd's AccumulateGrad
    adopt the tensor as `.grad` without a copy or an add (one elementwise launch per parameter otherwise).  Handed
    out once per step: a parameter used twice gets an ordinary temporary the second time and autograd adds it."""
    if param is None or not getattr(param, "_spe_grad_fresh", False) or getattr(param, "_spe_shared", False):
        return None             # (_spe_shared: used by several nodes of the graph - autograd sums their gradients itself, see dp._on_grad)
    param._spe_grad_fresh = False
    param._spe_handed = True          # explicit per-step flag (re-armed by the reducer's reset()): the view went to a kernel THIS step
    buf = param._spe_grad_buf
    return buf.view_as(buf)            # fresh alias: AccumulateGrad only steals a tensor nobody else references


ACC_ZERO_MAX = 16384       # == spe_amd.dp.GradAllReducer.ZERO_MAX: bucket views up to this size are zeroed by reducer.reset() every step


def _zeros_or(buf, n, device):
    """Destination of an ACCUMULATE-INTO gradient (bias / LayerNorm / LayerScale column sums: their kernels add partial sums onto
    what is there): the parameter's bucket view, or a zeroed temporary.  The reducer pre-zeroes only views of <= ACC_ZERO_MAX
    elements from the second step on (larger gradients are overwritten by their producers), so a larger accumulate-into view - a
    bias of a wide class / vocabulary head - is zeroed here, at hand-out time."""
    if buf is None:
        return zeros_small(n, device)
    v = buf.view(-1)
    if v.numel() > ACC_ZERO_MAX:
        v.zero_()
    return v


def _p(t):
    # a plain int: ctypes converts it for a c_void_p argtype itself (no Python-side c_void_p object per argument)
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _st():
    """Current HIP stream handle of the current device (the raw-handle query skips building a torch.cuda.Stream object).  Every
    wrapper fetches its launch stream here, so this is also where a change of stream or device is noticed (_guard_reduce_ws)."""
    cur = torch.cuda.current_device()
    st = _raw_stream(cur) if _raw_stream is not None else torch.cuda.current_stream().cuda_stream
    if st != _RWS_STREAM or cur != _RWS_DEVIDX:
        _guard_reduce_ws(cur, st)
    return st


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise lib.SpeLibraryError("spe_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.float32:
            raise TypeError(f"expected float32, got {t.dtype}")


TN_SMALL_TILES = True     # 64 x 64 tiles for weight gradients with few output tiles (module attributes, not environment knobs)
TN_SMALL_MAX = 256


def auto_splitk(M, N, K, batch):
    """Split-K factor for long contractions with few output tiles (the dW GEMMs).  Sized for 128x128 tiles
    (half the L2 re-reads of 64x64: 45 vs 74 us for fc1's dW) and ~2 workgroups per CU; the partial results go to
    private slabs that one column-sum launch adds up."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
    if tiles >= 256 or K < 1024:
        return 1
    # tiles * splits must FIT the 512 resident workgroup slots (2 per CU): 36 tiles x 15 = 540 leaves 28 workgroups for a
    # second, nearly empty round (fc1 / fc2 dW: 29 -> 24 us with 14 splits)
    sk = max(1, min(512 // tiles, K // 512, 16))
    if TN_SMALL_TILES and batch == 1 and tiles * sk < TN_SMALL_MAX:
        # few output tiles even at the largest split (a 384 x 384 weight: 9 tiles x 16 = 144 workgroups on 512 slots): 64 x 64 tiles
        # (spe_gemm_bf16tn picks them when 128-tiles x splits < 256) quadruple the tile count; the split is sized for THEM
        t64 = ((M + 63) // 64) * ((N + 63) // 64)
        sk = max(1, min(512 // t64, K // 512, 16))
    return sk


def gemm(A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, C2=None,
         batch0=1, batch1=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), alpha=1.0, act=0, splitk=1):
    """Raw strided (batched) GEMM on already-allocated tensors; returns C."""
    _call("spe_gemm_f32", _p(A), _p(B), _p(C), _p(bias), _p(C2), M, N, K, lda, ldb, ldc,
             int(transA), int(transB), batch0, batch1, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1],
             float(alpha), int(act), int(splitk), int(split_now()), _st())
    return C


# ---- nn.Linear ------------------------------------------------------------------------------
# Benchmark ("bf16") precision, large row counts: the three GEMMs of a Linear run on bf16 copies of their operands
# (csrc/gemm_bf16.hip) - the same roundings spe_gemm_f32 applies while staging, so the products are identical, with
# half the operand bytes and a deeper load pipeline.  Small GEMMs (decoder, heads) and the bf16x3 parity mode keep the
# fp32-operand kernel.
LINEAR16 = True              # module attributes (tools/error_budget.py and the tests flip them), not environment knobs
LINEAR16_MIN_ROWS = 128
_W16 = {}        # id(W) -> (weakref, version, epoch, W16 [N,K], W16T [K,N], W16lo (bf16 low part or IEEE fp16 copy, or None))
_W16_EPOCH = 0   # bumped by writers that bypass autograd's version counters (spe_amd.optim.FlatAdamW)


def weights_changed():
    """Invalidate the cached bf16 weight copies.  torch optimisers bump `Tensor._version`, which the cache checks; a
    kernel that updates parameters through raw pointers (FlatAdamW) must call this after every step."""
    global _W16_EPOCH
    _W16_EPOCH += 1



LINEAR_SMALL = True
LINEAR_SMALL_MAX_ROWS = 2048


def _lin_small_ok(R, N, K):
    """The one-launch Linear of csrc/linear_small.hip: row-major bf16 saves (DW_TN), 128 <= rows < 2048."""
    return (LINEAR_SMALL and DW_TN and LINEAR16 and _PRECISION != 1 and LINEAR16_MIN_ROWS <= R < LINEAR_SMALL_MAX_ROWS
            and N % 8 == 0 and K % 8 == 0)


def _lin16_ok(R, N, K):
    return LINEAR16 and _PRECISION != 1 and R >= LINEAR16_MIN_ROWS and N % 8 == 0 and K % 8 == 0


def cvt_bf16(x2, want=True, wantT=False, ldt=None, colsum_out=None, act_aux=None, act=0, out=None, ldo=None, out_lo=None, lo_f16=False):
    """bf16 (RNE) copies of a contiguous fp32 [R, C]: row-major [R, C] and/or the transpose [C, ldt] (zero padded);
    colsum_out (zeroed or running fp32 [C]) += column sums of x2 from the same pass.  out / ldo: write the row-major copy
    into a column block of a wider bf16 matrix (row stride ldo) instead of a fresh tensor.  out_lo (same layout as out):
    receives bf16(x - bf16(x)), the low part of a split operand."""
    _chk(x2)
    R, C = x2.shape
    if out is None:
        out = torch.empty((R, C), device=x2.device, dtype=torch.bfloat16) if want else None
        ldo = C
    outT = None
    if wantT:
        ldt = ldt or ((R + 63) // 64) * 64
        outT = torch.empty((C, ldt), device=x2.device, dtype=torch.bfloat16)
    if lo_f16:          # out_lo (an fp16 tensor) receives IEEE fp16(x): the operand of a single-term fp16 forward product
        assert colsum_out is None and act_aux is None and out_lo is not None and out_lo.dtype == torch.float16
        _call("spe_cvt_bf16_h", _p(x2), x2.stride(0), R, C, _p(out), _p(out_lo), ldo or C, _p(outT), ldt or 0, _st())
        return out, outT
    _call("spe_cvt_bf16", _p(x2), x2.stride(0), R, C, _p(out), _p(out_lo), ldo or C, _p(outT), ldt or 0, _p(colsum_out), _p(act_aux),
          int(act), _st())
    return out, outT


_W16_TABLE = None      # (signature, device job table, njobs, total tiles) of the last batched refresh
_W16_REFRESHED = -1    # epoch of the last batched refresh
_W16_BATCH = True


def _refresh_weights16():
    """Re-convert every cached weight copy in ONE launch (spe_cvt_bf16_multi) into its existing buffers: after an
    optimizer step ~190 weights are stale at once and each conversion alone is launch-latency bound."""
    global _W16_TABLE, _W16_REFRESHED
    import numpy as np
    live = []
    for key, ent in list(_W16.items()):
        owner = ent[0]()
        if owner is None or not _owns(owner, key[0]):
            del _W16[key]
            continue
        live.append((key, owner, ent))
    _W16_REFRESHED = _W16_EPOCH
    if not live:
        return
    # the table is reused only while every entry still converts the same storage into the SAME output buffers: an entry re-created
    # under an unchanged key (a later model whose flat parameter buffer landed on the address of a freed one) has new buffers
    sig = tuple((k, e[3].data_ptr(), e[4].data_ptr(), e[5].data_ptr() if e[5] is not None else 0, e[5].dtype if e[5] is not None else None)
                for k, _, e in live)
    if _W16_TABLE is None or _W16_TABLE[0] != sig:
        rec = np.zeros(len(live), dtype=np.dtype([("x", "<u8"), ("out", "<u8"), ("outT", "<u8"), ("ldt", "<i8"), ("R", "<i4"),
                                                   ("C", "<i4"), ("tile0", "<i4"), ("tiles_c", "<i4"), ("out_lo", "<u8"), ("flags", "<i8")]))
        t0 = 0
        for i, ((ptr, R, C), _, e) in enumerate(live):
            tc = (C + 63) // 64
            rec[i] = (ptr, e[3].data_ptr(), e[4].data_ptr(), e[4].shape[1], R, C, t0, tc, e[5].data_ptr() if e[5] is not None else 0,
                      1 if (e[5] is not None and e[5].dtype == torch.float16) else 0)          # (an fp16 second copy: weight16(..., f16=True))
            t0 += tc * ((max(R, e[4].shape[1]) + 63) // 64)
        table = host_table(rec.view(np.uint8), live[0][2][3].device)
        _W16_TABLE = (sig, table, len(live), t0)
    _, table, n, tiles = _W16_TABLE
    _call("spe_cvt_bf16_multi", _p(table), n, tiles, _st())
    for key, owner, e in live:
        _W16[key] = (e[0], owner._version, _W16_EPOCH, e[3], e[4], e[5])


def _owns(owner, ptr):
    """The storage `owner` holds now still covers the address a cache entry was made for."""
    base = owner.data_ptr()
    return base <= ptr < base + max(1, owner.numel()) * owner.element_size()


def host_table(bytes_np, device):
    """Small host-built table -> device through a pinned staging buffer (no pageable-copy pipeline drain)."""
    pin = torch.from_numpy(bytes_np.copy()).pin_memory()
    dev = torch.empty(pin.shape, dtype=torch.uint8, device=device)
    dev.copy_(pin, non_blocking=True)
    dev._spe_pin = pin            # keep the staging buffer alive until the copy has certainly run
    return dev


def weight16(W, lo=False, f16=False):
    """(W16 [N,K], W16T [K,N]) of a contiguous 2-D weight (or 2-D view of one: the patch-embedding filter), cached until
    the optimizer (or a state-dict load) changes it; lo=True: (W16, W16T, W16lo [N,K]) with W16lo = bf16(W - W16), the low
    part of the split forward operand (precision mode bf16s) - or, f16=True, the IEEE fp16 copy of W (the operand of the
    single-term fp16 forward products of the backbone MLP; such a weight never needs its low part).  Cache key: (address, shape);
    an entry lives as long as the tensor that owns the storage.  The first lookup after weights_changed() refreshes every cached
    copy in one launch."""
    import weakref
    key = (W.data_ptr(), W.shape[0], W.shape[1])
    ent = _W16.get(key)
    if ent is not None:
        owner = ent[0]()
        if owner is not None and _owns(owner, key[0]):
            if _W16_BATCH and ent[2] != _W16_EPOCH and _W16_REFRESHED != _W16_EPOCH:
                _refresh_weights16()
                ent = _W16.get(key, ent)
            if ent[1] == owner._version and ent[2] == _W16_EPOCH and (not lo or (ent[5] is not None and (ent[5].dtype == torch.float16) == f16)):
                return (ent[3], ent[4], ent[5]) if lo else (ent[3], ent[4])
            if not lo and ent[5] is not None and ent[5].dtype == torch.float16:
                f16 = True          # a stale entry of an fp16-forward weight looked up by its backward: it stays one
    owner = W._base if W._base is not None else W
    want_lo = lo or f16 or _PRECISION == 2
    with torch.no_grad():
        W16lo = torch.empty(W.shape, device=W.device, dtype=torch.float16 if f16 else torch.bfloat16) if want_lo else None
        W16, W16T = cvt_bf16(W.detach(), True, True, ldt=W.shape[0], out_lo=W16lo, lo_f16=f16)
    if len(_W16) > 4096:
        _W16.clear()
    _W16[key] = (weakref.ref(owner), owner._version, _W16_EPOCH, W16, W16T, W16lo)
    return (W16, W16T, W16lo) if lo else (W16, W16T)


_WCAT = {}      # ((address, shape) per weight) -> (owner weakrefs, epoch, versions, Wcat16, Wcat16T, bcat, Wcat16lo)


def weightcat16(Ws, bs, lo=False):
    """bf16 copies of several [N_i, K] weights stacked along the output axis (and the transposed stack for the input gradient)
    plus the stacked fp32 biases - the operands of ONE GEMM that evaluates all the Linears sharing an input (ops.multi_linear);
    lo=True: also the stacked low parts (split forward operand).  Rebuilt from the cached per-weight copies when any weight
    changed (three concatenations).  Keyed like weight16 on (address, shape) with weak references to the owning tensors: an
    `id()` can be reused by a later model in the same process."""
    import weakref
    key = tuple((W.data_ptr(), tuple(W.shape)) for W in Ws)
    vers = tuple(W._version for W in Ws) + tuple(b._version for b in bs)
    ent = _WCAT.get(key)
    if ent is not None and ent[1] == _W16_EPOCH and ent[2] == vers and (ent[6] is not None or not lo):
        owners = [r() for r in ent[0]]
        if all(o is not None and _owns(o, W.data_ptr()) for o, W in zip(owners, Ws)) and ent[3].shape == (sum(W.shape[0] for W in Ws), Ws[0].shape[1]):
            return (ent[3], ent[4], ent[5], ent[6]) if lo else (ent[3], ent[4], ent[5])
    with torch.no_grad():
        trip = [weight16(W, lo=lo) for W in Ws]
        Wc = torch.cat([t_[0] for t_ in trip], 0)
        WcT = torch.cat([t_[1] for t_ in trip], 1)
        Wclo = torch.cat([t_[2] for t_ in trip], 0) if lo else None
        bc = torch.cat([b.detach() for b in bs])
    if len(_WCAT) > 64:
        _WCAT.clear()
    _WCAT[key] = ([weakref.ref(W._base if W._base is not None else W) for W in Ws], _W16_EPOCH, vers, Wc, WcT, bc, Wclo)
    return (Wc, WcT, bc, Wclo) if lo else (Wc, WcT, bc)


def gemm_splitk_into(A, B, out, M, N, K, lda, ldb, ldc, transA, transB, batch0, batch1, sA, sB, sC, alpha=1.0):
    """Batched GEMM with a long contraction and few output tiles (decoder cross-attention P.V / dS.K: 64 workgroups
    looping 130 K-steps): split K into private slabs and sum them into the contiguous `out` whose layout the batch
    strides sC describe.  Falls back to the plain launch when splitting does not pay."""
    sk = auto_splitk(M, N, K, batch0 * batch1)
    if sk <= 1:
        return gemm(A, B, out, M, N, K, lda, ldb, ldc, transA, transB, batch0=batch0, batch1=batch1, sA=sA, sB=sB, sC=sC, alpha=alpha)
    extent = (((batch0 - 1) * sC[0] + (batch1 - 1) * sC[1] + (M - 1) * ldc + N) + 3) & ~3
    assert out.is_contiguous() and extent <= out.numel() + 3
    ws = torch.empty((sk, extent), device=out.device, dtype=torch.float32)
    gemm(A, B, ws, M, N, K, lda, ldb, ldc, transA, transB, batch0=batch0, batch1=batch1, sA=sA, sB=sB, sC=sC, alpha=alpha, splitk=-sk)
    out.zero_()
    n = out.numel()
    _call("spe_colsum", _p(ws), _p(out), sk, n, extent, 1, _st())
    return out


def gemm16(A16, B16, C, M, N, K, lda, ldb, ldc, bias=None, C2=None, alpha=1.0, act=0, splitk=1, Alo=None, Blo=None):
    """C = act(alpha * A16 @ B16.T + bias) on bf16 operands (both k-contiguous); splitk < 0: slabs.  Alo / Blo (both or
    neither): the low parts of split operands - the product becomes A B + Alo B + A Blo."""
    _call("spe_gemm_bf16nt", _p(A16), _p(B16), _p(Alo), _p(Blo), _p(C), _p(bias), _p(C2), M, N, K, lda, ldb, ldc, float(alpha), int(act),
          int(splitk), _st())
    return C


def cvt_f16(x2, out=None):
    """IEEE fp16 copy (saturating) of a contiguous fp32 [R, C], C % 4 == 0."""
    R, C = x2.shape
    if out is None:
        out = torch.empty((R, C), device=x2.device, dtype=torch.float16)
    _call("spe_cvt_f16", _p(x2), x2.stride(0), R, C, _p(out), out.stride(0), _st())
    return out


_WCATH = {}     # fp16 stacks of weights sharing an input: key -> (owner weakrefs, epoch, versions, Wcat fp16 [n*N, K], bcat fp32)


def weightcat_f16(Ws, bs):
    """IEEE fp16 copy of several [N, K] weights stacked along the output axis + the stacked fp32 biases: the single-term fp16 operand
    of the decoder's memory-side projection GEMM (spe_gemm_bf16nt, act bit 8).  Rebuilt when a weight changed (one concatenation +
    one conversion launch)."""
    import weakref
    key = tuple((W.data_ptr(), tuple(W.shape)) for W in Ws)
    vers = tuple(W._version for W in Ws) + tuple(b._version for b in bs)
    ent = _WCATH.get(key)
    if ent is not None and ent[1] == _W16_EPOCH and ent[2] == vers:
        owners = [r() for r in ent[0]]
        if all(o is not None and _owns(o, W.data_ptr()) for o, W in zip(owners, Ws)):
            return ent[3], ent[4]
    with torch.no_grad():
        Wh = cvt_f16(torch.cat([W.detach() for W in Ws], 0))
        bc = torch.cat([b.detach() for b in bs])
    if len(_WCATH) > 64:
        _WCATH.clear()
    _WCATH[key] = ([weakref.ref(W._base if W._base is not None else W) for W in Ws], _W16_EPOCH, vers, Wh, bc)
    return Wh, bc


def kv_frags(ym16, yp16, L, B, S, H, dh, train):
    """fp16 outputs of the stacked memory-side projections -> (Kf, V16, K16, Vf) fragment stacks [L, B, H, nt, ...] (csrc/decoder_kv.hip);
    K16 / Vf (the backward's bf16 operands) only when `train`."""
    dev = ym16.device
    nt, dk = (S + 15) // 16, 2 * dh
    Kf = torch.empty((L, B, H, nt, (dk + 31) // 32, 64, 8), device=dev, dtype=torch.float16)
    V16 = torch.empty((L, B, H, nt, (dh + 15) // 16, 64, 4), device=dev, dtype=torch.float16)
    K16 = torch.empty((L, B, H, nt, (dk + 15) // 16, 64, 4), device=dev, dtype=torch.bfloat16) if train else None
    Vf = torch.empty((L, B, H, nt, (dh + 31) // 32, 64, 8), device=dev, dtype=torch.bfloat16) if train else None
    _call("spe_kv_frags", _p(ym16), ym16.stride(0), _p(yp16), yp16.stride(0), _p(Kf), _p(V16), _p(K16), _p(Vf), L, B, S, H, dh, _st())
    return Kf, V16, K16, Vf


def kv_grad_scatter(dk, dv, dYm, dYp, layer, B, S, H, dh):
    _call("spe_kv_grad_scatter", _p(dk), _p(dv), _p(dYm), dYm.stride(0), _p(dYp), dYp.stride(0), int(layer), B, S, H, dh, _st())


def colsum_bf16_blocks(x16, blkC, outs):
    """Column sums of the bf16 matrix x16 [R, n*blkC], block i -> outs[i] (fp32 [blkC], overwritten)."""
    n = len(outs)
    ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    _call("spe_colsum_bf16_blocks", _p(x16), x16.stride(0), x16.shape[0], n, int(blkC), ptrs, 0, _st())


def gemm16_tn(A16, B16, C, M, N, R, lda, ldb, ldc, alpha=1.0, splitk=1):
    """C[M,N] = alpha * A16[:R,:M].T @ B16[:R,:N] on row-major bf16 operands (spe_gemm_bf16tn); splitk < 0: slabs."""
    _call("spe_gemm_bf16tn", _p(A16), _p(B16), _p(C), M, N, R, lda, ldb, ldc, float(alpha), int(splitk), _st())
    return C


def gemm16_ex(A16, B16, M, N, K, lda, ldb, bias=None, C=None, C2=None, out16=None, out16T=None, colsum=None, aux=None,
              alpha=1.0, act=0, res=None, rgamma=None, Alo=None, Blo=None, out16lo=None, drop=None, sscale=None, rps=1, op_f16=False):
    """spe_gemm_bf16nt_ex: v = alpha * A16 @ B16.T + bias; C2 = v; v = act(v) or v * act'(aux); optional fp32 C [M,N], bf16
    out16 [M,N], bf16 transposed out16T [N, ldt] (zero padded), colsum [N] += column sums.  All fp32 tensors have ld = N.
    Alo / Blo: low parts of split operands; out16lo [M,N]: bf16(v - out16), the low part of the result for the next split GEMM.
    C2 / aux may be fp16 tensors (the saved pre-activation of the fused MLP): flagged to the library by their dtype."""
    half_flags = (1 if (C2 is not None and C2.dtype == torch.float16) else 0) | (2 if (aux is not None and aux.dtype == torch.float16) else 0)
    if op_f16:          # A16 / B16 hold IEEE fp16 (single-term product); an fp16 out16lo tensor receives fp16(v), the next fp16 product's operand
        assert A16.dtype == torch.float16 and B16.dtype == torch.float16 and Alo is None and Blo is None
        half_flags |= 4 | (8 if out16lo is not None else 0)
        assert out16lo is None or out16lo.dtype == torch.float16
    if (drop is not None and drop[0] > 0) or sscale is not None:
        # drop = (p, seed, offset): dropout after the activation / derivative; sscale [B] (+ rps rows per sample): DropPath scale on the residual
        pd, sd, of = drop if (drop is not None and drop[0] > 0) else (0.0, 0, 0)
        _call("spe_gemm_bf16nt_exd", _p(A16), _p(B16), _p(Alo), _p(Blo), _p(C), _p(bias), _p(C2), _p(out16), _p(out16lo), N, _p(out16T),
              out16T.shape[1] if out16T is not None else 0, _p(colsum), _p(aux), _p(res), _p(rgamma), M, N, K, lda, ldb, N,
              float(alpha), int(act), half_flags, float(pd), int(sd), int(of), _p(sscale), int(rps), _st())
        return
    _call("spe_gemm_bf16nt_ex", _p(A16), _p(B16), _p(Alo), _p(Blo), _p(C), _p(bias), _p(C2), _p(out16), _p(out16lo), N, _p(out16T),
          out16T.shape[1] if out16T is not None else 0, _p(colsum), _p(aux), _p(res), _p(rgamma), M, N, K, lda, ldb, N,
          float(alpha), int(act), half_flags, _st())


def mlp16_ok(R, K, Hd, N):
    """The fused bf16 MLP path (spe_amd.ops.mlp_gelu): both Linears on the bf16-copy GEMMs."""
    return _lin16_ok(R, Hd, K) and _lin16_ok(R, N, Hd)


# Weight gradients on ROW-MAJOR bf16 operands (spe_gemm_bf16tn: LDS transpose reads): no producer writes a transposed bf16
# copy any more - not the activation conversions (x16T), not the backward conversions (dy16T), not the GEMM epilogues.
DW_TN = True


def layerscale_residual_bwd16(dout2, y2, gamma, Rp, db_out=None, dg_out=None, want_rowmajor=True, want_T=True, drop=None, sscale=None, rps=1):
    """Backward of out = x + gamma * y when y is the output of a Linear on the bf16-copy GEMMs: -> (dy16 [R,C], dy16T
    [C,Rp], db [C], dgamma [C]); dy = gamma * dout exists only as those bf16 operands."""
    R, C = dout2.shape
    dev = dout2.device
    dy16 = torch.empty((R, C), device=dev, dtype=torch.bfloat16) if want_rowmajor else None
    dy16T = torch.empty((C, Rp), device=dev, dtype=torch.bfloat16) if want_T else None
    db = _zeros_or(db_out, C, dev)
    dg = _zeros_or(dg_out, C, dev)
    if (drop is not None and drop[0] > 0) or sscale is not None:       # the forward was x + s_b * gamma * dropout(y)
        pd, sd, of = drop if (drop is not None and drop[0] > 0) else (0.0, 0, 0)
        _call("spe_layerscale_residual_bwd16d", _p(dout2), _p(y2), int(y2.dtype == torch.float16), _p(gamma), _p(dy16), _p(dy16T), Rp, _p(db),
              _p(dg), R, C, float(pd), int(sd), int(of), _p(sscale), int(rps), _st())
        return dy16, dy16T, db, dg
    _call("spe_layerscale_residual_bwd16", _p(dout2), _p(y2), int(y2.dtype == torch.float16), _p(gamma), _p(dy16), _p(dy16T), Rp, _p(db), _p(dg),
          R, C, _st())
    return dy16, dy16T, db, dg


def linear_res_fwd(x2, W, b, res, gamma, save=True, src=None, drop=None, sscale=None, rps=1):
    """out = res + gamma * (x2 @ W.T + b) on the bf16-copy GEMM with the residual in its epilogue.  -> (out, (x16T, y))."""
    R, K = x2.shape
    N = W.shape[0]
    dev = x2.device
    sp = split_fwd()
    x16, x16T, x16lo = act16(x2, save and not DW_TN, src, want_lo=sp)
    out = torch.empty((R, N), device=dev, dtype=torch.float32)
    # the branch output y is kept for the LayerScale gamma gradient only (dgamma = sum dout * y): fp16, like the MLP's pre-activation
    y = torch.empty((R, N), device=dev, dtype=torch.float16 if MLP_PRE_F16 else torch.float32) if save else None
    Wt = weight16(W, lo=sp)
    gemm16_ex(x16, Wt[0], R, N, K, K, K, bias=b, C=out, C2=y, res=res, rgamma=gamma, Alo=x16lo, Blo=Wt[2] if sp else None,
              drop=drop, sscale=sscale, rps=rps)
    return out, ((x16 if DW_TN else x16T), y)


def linear_res_bwd(dout2, saved, W, gamma, need_dx=True, grad_bufs=(None, None, None), drop=None, sscale=None, rps=1, pre=None):
    """Backward of linear_res_fwd: (dx, dW, db, dgamma); gamma * dout only exists as the bf16 operands of the two GEMMs.
    pre = (dy16, db, dgamma): the LayerScale part was already taken by the LayerNorm backward that produced dout (layernorm_bwd ls=)."""
    xs, y = saved
    R, N = dout2.shape
    K = W.shape[1]
    gW, gb, gg = grad_bufs
    if _is_rowmajor_save(xs, R):
        Rp = ((R + 63) // 64) * 64
        if pre is not None:
            dy16, db, dg = pre
        else:
            dy16, _, db, dg = layerscale_residual_bwd16(dout2, y, gamma, Rp, db_out=gb, dg_out=gg, want_rowmajor=True, want_T=False,
                                                        drop=drop, sscale=sscale, rps=rps)
        dW = _dw16_tn(dy16, xs, N, K, R, gW)
    else:
        if pre is not None:         # the sums are already in db / dgamma: running the LayerScale backward again would count them twice
            raise RuntimeError("spe_amd.kernels.linear_res_bwd: a LayerNorm backward took this node's LayerScale part, but its save is not row-major")
        Rp = xs.shape[1]
        dy16, dy16T, db, dg = layerscale_residual_bwd16(dout2, y, gamma, Rp, db_out=gb, dg_out=gg, want_rowmajor=need_dx,
                                                        drop=drop, sscale=sscale, rps=rps)
        dW = _dw16(dy16T, xs, N, K, Rp, gW)
    dx = None
    if need_dx:
        dx = torch.empty((R, K), device=dout2.device, dtype=torch.float32)
        gemm16(dy16, weight16(W)[1], dx, R, K, N, N, N, K)
    return dx, dW, db, dg


# Saves that only the backward reads (the MLP's pre-activation for gelu'(.), the branch outputs for the LayerScale gamma gradients) as
# IEEE fp16 instead of fp32: half the bytes, but from the accumulator layout a 16-bit tile leaves as 32-B row pieces - the fp32 stores are
# faster than the fp16 ones they replaced (fc1 + GELU isolated: 74.6 vs 80.9 us; step 54.9 -> 54.4 ms in same-box A/B, round 4), and staging
# the tile through LDS costs a workgroup per CU.  Off by default since; 1.1 GB more saved activations at cfg2.
MLP_PRE_F16 = False      # module attribute (tests/test_kernels_gpu.py runs both settings), not an environment knob


# Round 5: the FORWARD products of the backbone MLP (fc1, fc2: two thirds of the forward Linear FLOPs of a block) in precision mode bf16s run
# on single-term IEEE fp16 operands instead of split bf16 pairs - one matrix instruction and half the operand bytes per product instead of
# three and two.  Error budget (tools/error_budget.py mlp_fp16, profiles/r05_error_budget.jsonl): fp16 operands in fc1 + fc2 alone move the
# worst weighted loss key by 1.3e-4 (cfg2 full depth) / 2.6e-4 (cfg5 full depth), pred_logits by 8.6e-5 - the other families (qkv 4.2e-4,
# proj 6.4e-4 on their worst key) keep the split.  The producers emit the fp16 copy in the slot of the low part (LayerNorm, the fc1
# epilogue, the weight copies); the backward still reads the bf16 copies.  Module attribute (tests run both settings), not an environment knob.
MLP_F16 = True


def mlp_f16_ok(R, K, Hd, N):
    # mirrors spe_nt2_dispatch (csrc/gemm_nt2.hip): the fp16-operand extended epilogue exists only there - M >= 2048, contraction % 64 == 0 and
    # >= 128, >= 64 output columns, for BOTH products (fc1: K -> Hd, fc2: Hd -> N); anything else takes the split-bf16 path
    return (MLP_F16 and split_fwd() and R >= 2048 and K % 64 == 0 and K >= 128 and Hd % 64 == 0 and Hd >= 128 and N % 8 == 0 and N >= 64)


def mlp_gelu_fwd(x2, W1, b1, W2, b2, res=None, gamma=None, save=True, src=None, drop1=None, drop2=None, sscale=None, rps=1):
    """y = fc2(gelu(fc1(x2))) with every intermediate that the next GEMM needs emitted as bf16 by the producing GEMM's
    epilogue: fc1 writes the fp32 pre-activation (for the backward) and the bf16 activation h16 / h16T, never the fp32
    activation.  -> (y [R,N] fp32, saved = (x16T, pre, h16T))."""
    R, K = x2.shape
    Hd, N = W1.shape[0], W2.shape[0]
    dev = x2.device
    # save = False (no gradient wanted: inference): none of the tensors that only the backward reads is produced
    sp = split_fwd()                         # bf16s forward: both products on (hi, lo) operand pairs, fc1 emits gelu(pre) as a pair
    if res is not None and DW_TN and mlp_f16_ok(R, K, Hd, N):
        # ... or on single-term fp16 operands: x as (bf16 for the backward, fp16), fc1 emits gelu(pre) as (bf16 for the backward, fp16 for fc2)
        x16, _, xh = act16(x2, False, src, want_lo=True, lo_f16=True)
        pre = torch.empty((R, Hd), device=dev, dtype=torch.float16 if MLP_PRE_F16 else torch.float32) if save else None
        h16 = torch.empty((R, Hd), device=dev, dtype=torch.bfloat16)
        hh = torch.empty((R, Hd), device=dev, dtype=torch.float16)
        W1h, W2h = weight16(W1, lo=True, f16=True)[2], weight16(W2, lo=True, f16=True)[2]
        gemm16_ex(xh, W1h, R, Hd, K, K, K, bias=b1, C2=pre, out16=h16, act=2, out16lo=hh, drop=drop1, op_f16=True)
        out = torch.empty((R, N), device=dev, dtype=torch.float32)
        y = torch.empty((R, N), device=dev, dtype=torch.float16 if MLP_PRE_F16 else torch.float32) if save else None
        gemm16_ex(hh, W2h, R, N, Hd, Hd, Hd, bias=b2, C=out, C2=y, res=res, rgamma=gamma, drop=drop2, sscale=sscale, rps=rps, op_f16=True)
        return out, (x16, pre, h16, y)
    x16, x16T, x16lo = act16(x2, save and not DW_TN, src, want_lo=sp)
    Rp = ((R + 63) // 64) * 64
    # the pre-activation is kept for gelu'(.) of the backward only: fp16 (11 significant bits: the derivative is exact to ~3e-4,
    # an order below the bf16 operand rounding of the backward products) - half the bytes of the largest activation of the block
    pre = torch.empty((R, Hd), device=dev, dtype=torch.float16 if MLP_PRE_F16 else torch.float32) if save else None
    h16 = torch.empty((R, Hd), device=dev, dtype=torch.bfloat16)
    h16lo = torch.empty((R, Hd), device=dev, dtype=torch.bfloat16) if sp else None
    h16T = torch.empty((Hd, Rp), device=dev, dtype=torch.bfloat16) if (save and not DW_TN) else None
    W1t, W2t = weight16(W1, lo=sp), weight16(W2, lo=sp)
    W1lo, W2lo = (W1t[2], W2t[2]) if sp else (None, None)
    # drop1 = (p, seed, offset): timm Mlp's dropout after the activation - h16 then holds dropout(gelu(pre))
    gemm16_ex(x16, W1t[0], R, Hd, K, K, K, bias=b1, C2=pre, out16=h16, out16T=h16T, act=2, Alo=x16lo, Blo=W1lo, out16lo=h16lo, drop=drop1)
    y = torch.empty((R, N), device=dev, dtype=torch.float32)
    if DW_TN:
        x16T, h16T = x16, h16            # what the backward gets: the row-major copies
    if res is None:
        gemm16(h16, W2t[0], y, R, N, Hd, Hd, Hd, N, bias=b2, Alo=h16lo, Blo=W2lo)
        return y, (x16T, pre, h16T)
    # LayerScale residual in the fc2 epilogue: out = res + gamma * y ; y is kept for the gamma gradient
    out = torch.empty((R, N), device=dev, dtype=torch.float32)
    if MLP_PRE_F16:
        y = torch.empty((R, N), device=dev, dtype=torch.float16)        # kept for the gamma gradient only
    gemm16_ex(h16, W2t[0], R, N, Hd, Hd, Hd, bias=b2, C=out, C2=y if save else None, res=res, rgamma=gamma, Alo=h16lo, Blo=W2lo,
              drop=drop2, sscale=sscale, rps=rps)
    return out, (x16T, pre, h16T, y)


def _dw16(dy16T, x16T, N, K, Rp, dW_out):
    """dW [N,K] = dy16T [N,Rp] @ x16T[K,Rp]^T, split-K slabs summed into dW_out (zeroed bucket view) when given."""
    dev = dy16T.device
    sk = min(auto_splitk(N, K, Rp, 1), Rp // 64)
    if sk > 1:
        ws = torch.empty((sk, N * K), device=dev, dtype=torch.float32)
        gemm16(dy16T, x16T, ws, N, K, Rp, Rp, Rp, K, splitk=-sk)
        return colsum(ws, out=None if dW_out is None else dW_out.view(-1), accumulate=dW_out is None).view(N, K)
    dW = dW_out if dW_out is not None else torch.empty((N, K), device=dev, dtype=torch.float32)
    gemm16(dy16T, x16T, dW, N, K, Rp, Rp, Rp, K)
    return dW


def _is_rowmajor_save(xs, R):
    """Saved bf16 activations are row-major [R, K] (DW_TN, fixed for the life of the process) or the padded transpose [K, Rp]."""
    return DW_TN and xs.dtype == torch.bfloat16


def _dw16_tn(dy16, x16, N, K, R, dW_out, lda=None):
    """dW [N,K] = dy16 [R,N]^T @ x16 [R,K] on row-major operands; split over the rows into slabs summed into dW_out.
    lda: row stride of dy16 when it is a column block of a wider matrix."""
    dev = dy16.device
    lda = N if lda is None else lda
    sk = min(auto_splitk(N, K, R, 1), max(1, R // 64))
    if sk > 1:
        ws = torch.empty((sk, N * K), device=dev, dtype=torch.float32)
        gemm16_tn(dy16, x16, ws, N, K, R, lda, K, K, splitk=-sk)
        return colsum(ws, out=None if dW_out is None else dW_out.view(-1), accumulate=dW_out is None).view(N, K)
    dW = dW_out if dW_out is not None else torch.empty((N, K), device=dev, dtype=torch.float32)
    gemm16_tn(dy16, x16, dW, N, K, R, lda, K, K)
    return dW


def mlp_gelu_bwd(dy2, saved, W1, W2, need_dx=True, grad_bufs=(None, None, None, None), gamma=None, dg_out=None,
                 drop1=None, drop2=None, sscale=None, rps=1, ls_pre=None):
    """Backward of mlp_gelu_fwd.  dy2 [R,N] fp32.  -> (dx, dW1, db1, dW2, db2).  The gradient w.r.t. the pre-activation
    exists only as the bf16 copies (row-major for dx, transposed for dW1) written by the dh GEMM's epilogue, which also
    applies gelu' and accumulates db1.  grad_bufs: zeroed bucket views for (dW1, db1, dW2, db2) or None."""
    x16T, pre, h16T = saved[:3]
    R, N = dy2.shape
    Hd, K = W1.shape
    tn = _is_rowmajor_save(x16T, R)
    Rp = ((R + 63) // 64) * 64 if tn else x16T.shape[1]
    dev = dy2.device
    gW1, gb1, gW2, gb2 = grad_bufs
    dg = None
    if ls_pre is not None and (gamma is None or not tn):      # never fall through: the LayerNorm backward already added this node's sums
        raise RuntimeError("spe_amd.kernels.mlp_gelu_bwd: a LayerNorm backward took this node's LayerScale part, but the node cannot consume it")
    if gamma is not None and ls_pre is not None and tn:      # the LayerScale part came with the LayerNorm backward that produced dy2 (layernorm_bwd ls=)
        dy16, db2, dg = ls_pre
        dy16T = None
    elif gamma is not None:        # residual form: dy2 is d(out); the branch gradient gamma * dout only exists in bf16
        dy16, dy16T, db2, dg = layerscale_residual_bwd16(dy2, saved[3], gamma, Rp, db_out=gb2, dg_out=dg_out, want_T=not tn,
                                                         drop=drop2, sscale=sscale, rps=rps)
    else:
        db2 = _zeros_or(gb2, N, dev)
        dy16, dy16T = cvt_bf16(dy2, True, not tn, ldt=Rp, colsum_out=db2)
    dW2 = _dw16_tn(dy16, h16T, N, Hd, R, gW2) if tn else _dw16(dy16T, h16T, N, Hd, Rp, gW2)
    # dpre = (dy @ W2) * gelu'(pre): bf16 only
    db1 = _zeros_or(gb1, Hd, dev)
    dp16 = torch.empty((R, Hd), device=dev, dtype=torch.bfloat16) if (need_dx or tn) else None
    dp16T = torch.empty((Hd, Rp), device=dev, dtype=torch.bfloat16) if not tn else None
    gemm16_ex(dy16, weight16(W2)[1], R, Hd, N, N, N, out16=dp16, out16T=dp16T, colsum=db1, aux=pre, act=2, drop=drop1)      # ... times the mask of drop1
    dW1 = _dw16_tn(dp16, x16T, Hd, K, R, gW1) if tn else _dw16(dp16T, x16T, Hd, K, Rp, gW1)
    dx = None
    if need_dx:
        dx = torch.empty((R, K), device=dev, dtype=torch.float32)
        gemm16(dp16, weight16(W1)[1], dx, R, K, Hd, Hd, Hd, K)
    if gamma is not None:
        return dx, dW1, db1, dW2, db2, dg
    return dx, dW1, db1, dW2, db2


def act16(x2, wantT, src=None, want_lo=False, lo_f16=False):
    """bf16 copies (row-major, transposed when wantT, and the low part of the split operand when want_lo) of an activation
    [R,K] -> (x16, x16T, x16lo).  `src`: the tensor object the caller holds (x2 is a reshape of it) - the copies are
    remembered ON that object (attribute, checked against its version counter), so an activation that feeds several Linears
    (the decoder's memory, positional embedding, tgt, query_pos; q / k / v of the class attention) is converted once; they die
    with the tensor."""
    if src is not None:
        ent = getattr(src, "_spe16", None)
        if (ent is not None and ent[0] == src._version and ent[1].shape == x2.shape and (ent[2] is not None or not wantT)
                and (not want_lo or (ent[3] is not None and (ent[3].dtype == torch.float16) == lo_f16))):
            return ent[1], ent[2], (ent[3] if want_lo else None)
    # lo_f16: the second copy is IEEE fp16(x) - the operand of a single-term fp16 product - instead of the low part of the split
    x16lo = torch.empty(x2.shape, device=x2.device, dtype=torch.float16 if lo_f16 else torch.bfloat16) if want_lo else None
    x16, x16T = cvt_bf16(x2, True, wantT, out_lo=x16lo, lo_f16=lo_f16 and want_lo)
    if src is not None:
        src._spe16 = (src._version, x16, x16T, x16lo)
    return x16, x16T, x16lo


def linear_fwd(x2, W, b, act=0, want_pre=False, save_for_dw=True, src=None):
    """y = act(x2 @ W.T + b); x2 [R,K] contiguous, W [N,K].  -> (y, pre-activation or None, xsave): xsave is what
    linear_bwd needs of x - x2 itself, or on the bf16 path the padded bf16 transpose x16T [K, Rp]."""
    _chk(x2, W, b)
    R, K = x2.shape
    N = W.shape[0]
    y = torch.empty((R, N), device=x2.device, dtype=torch.float32)
    pre = torch.empty_like(y) if want_pre else None
    if _lin_small_ok(R, N, K) and W.is_contiguous():
        # a few hundred rows (decoder / heads): one launch straight from the fp32 activations (csrc/linear_small.hip); it also
        # leaves bf16(x) behind for the backward unless the activation already carries one
        sp = split_fwd()
        Wt = weight16(W, lo=sp)
        ent = getattr(src, "_spe16", None) if src is not None else None
        x16 = ent[1] if (ent is not None and ent[0] == src._version and ent[1].shape == x2.shape) else None
        x16_out = None
        if x16 is None and save_for_dw:
            x16 = x16_out = torch.empty((R, K), device=x2.device, dtype=torch.bfloat16)
            if src is not None:
                src._spe16 = (src._version, x16, None, None)
        _call("spe_linear_small_fwd", _p(x2), x2.stride(0), _p(Wt[0]), _p(Wt[2] if sp else None), _p(b), _p(y), _p(pre), _p(x16_out),
              R, N, K, N, int(act), _st())
        return y, pre, (x16 if save_for_dw else x2)
    if _lin16_ok(R, N, K) and W.is_contiguous():
        sp = split_fwd()
        x16, x16T, x16lo = act16(x2, save_for_dw and not DW_TN, src, want_lo=sp)
        Wt = weight16(W, lo=sp)
        gemm16(x16, Wt[0], y, R, N, K, K, K, N, bias=b, C2=pre, act=act, Alo=x16lo, Blo=Wt[2] if sp else None)
        return y, pre, ((x16 if DW_TN else x16T) if save_for_dw else x2)
    gemm(x2, W, y, R, N, K, K, K, N, False, True, bias=b, C2=pre, act=act)
    return y, pre, x2


LINEAR_GROUP = True


def linear_group_ok(R, Ws, bs):
    """Several Linears of one shape on one input of a few hundred rows: the one-launch group form of csrc/linear_small.hip."""
    N, K = Ws[0].shape
    return (LINEAR_GROUP and 2 <= len(Ws) <= 16 and _lin_small_ok(R, N, K) and N % 128 == 0
            and all(W.shape == Ws[0].shape and W.is_contiguous() for W in Ws) and all(b is not None for b in bs))


def _ptr_array(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*[(p if p else None) for p in ptrs])


def linear_group_fwd(x2, Ws, bs, src=None, adds=None):
    """[x2 W_i^T + b_i (+ adds[i])] as n separate contiguous [R, N] tensors from ONE launch; -> (ys, x16 save for the backward).  adds: list of
    contiguous fp32 [R, N] tensors or None entries."""
    _chk(x2, *Ws)
    R, K = x2.shape
    N = Ws[0].shape[0]
    sp = split_fwd()
    trip = [weight16(W, lo=sp) for W in Ws]
    ent = getattr(src, "_spe16", None) if src is not None else None
    x16 = ent[1] if (ent is not None and ent[0] == src._version and ent[1].shape == x2.shape) else None
    x16_out = None
    if x16 is None:
        x16 = x16_out = torch.empty((R, K), device=x2.device, dtype=torch.bfloat16)
        if src is not None:
            src._spe16 = (src._version, x16, None, None)
    ys = [torch.empty((R, N), device=x2.device, dtype=torch.float32) for _ in Ws]
    _call("spe_linear_small_group_fwd", _p(x2), x2.stride(0), _ptr_array([t_[0].data_ptr() for t_ in trip]),
          _ptr_array([t_[2].data_ptr() for t_ in trip]) if sp else None, _ptr_array([b.data_ptr() for b in bs]),
          _ptr_array([0 if a is None else a.data_ptr() for a in adds]) if adds is not None else None,
          _ptr_array([y.data_ptr() for y in ys]), _p(x16_out), R, len(Ws), N, K, _st())
    return ys, x16


def linear_group_bwd(dys, x16, Ws, need_dx, dW_outs, db_outs):
    """dys: list of contiguous fp32 [R, N] or None.  -> (dx or None, dWs, dbs); dW_outs / db_outs: bucket views (or None: fresh tensors);
    entries of outputs without a gradient stay None."""
    n = len(Ws)
    N, K = Ws[0].shape
    R = x16.shape[0]
    dev = x16.device
    dx = torch.empty((R, K), device=dev, dtype=torch.float32) if need_dx else None
    dWs, dbs = [], []
    for i in range(n):
        if dys[i] is None:
            dWs.append(None); dbs.append(None)
            continue
        dWs.append(dW_outs[i] if dW_outs[i] is not None else torch.empty((N, K), device=dev, dtype=torch.float32))
        dbs.append(db_outs[i].view(-1) if db_outs[i] is not None else torch.empty((N,), device=dev, dtype=torch.float32))
    _call("spe_linear_small_group_bwd", _ptr_array([0 if d is None else d.data_ptr() for d in dys]), _p(x16),
          _ptr_array([weight16(W)[1].data_ptr() for W in Ws]) if need_dx else None, _p(dx),
          _ptr_array([0 if w is None else w.data_ptr() for w in dWs]), _ptr_array([0 if b is None else b.data_ptr() for b in dbs]),
          R, n, N, K, _st())
    return dx, dWs, dbs


def linear_bwd(dy2, xsave, W, need_dx=True, need_dw=True, need_db=True, dW_out=None, db_out=None, act=0, act_aux=None):
    """dx = dy @ W ; dW = dy.T @ x ; db = colsum(dy).  xsave: third result of linear_fwd.
    dW_out / db_out: zeroed destination buffers (grad_buffer).  act/act_aux: dy is the gradient w.r.t. the OUTPUT of
    a fused activation (1 ReLU: aux = output, 2 GELU: aux = pre-activation); its backward is applied on the fly."""
    _chk(dy2, W)
    R, N = dy2.shape
    K = W.shape[1]
    dx = dW = db = None
    x16 = xsave.dtype == torch.bfloat16
    if (x16 or not need_dw) and _lin_small_ok(R, N, K) and W.is_contiguous() and (not x16 or _is_rowmajor_save(xsave, R)):
        # one launch: dx, dW and db (csrc/linear_small.hip); dW / db are overwritten, so the bucket views need no zeroing
        dev = dy2.device
        if need_dx:
            dx = torch.empty((R, K), device=dev, dtype=torch.float32)
        if need_dw:
            dW = dW_out if dW_out is not None else torch.empty((N, K), device=dev, dtype=torch.float32)
        if need_db:
            db = db_out.view(-1) if db_out is not None else torch.empty((N,), device=dev, dtype=torch.float32)
        _call("spe_linear_small_bwd", _p(dy2), _p(act_aux if act else None), int(act), _p(xsave if need_dw else None),
              _p(weight16(W)[1] if need_dx else None), _p(dx), _p(dW), _p(db), R, N, K, _st())
        return dx, dW, db
    if (x16 or not need_dw) and _lin16_ok(R, N, K) and W.is_contiguous():
        tn = x16 and _is_rowmajor_save(xsave, R)
        Rp = None if (tn or not x16) else xsave.shape[1]
        if need_db:                     # the bias gradient rides on the conversion pass over dy
            db = _zeros_or(db_out, N, dy2.device)
        dy16, dy16T = cvt_bf16(dy2, need_dx or (need_dw and tn), need_dw and x16 and not tn, ldt=Rp, colsum_out=db,
                               act_aux=act_aux if act else None, act=act)
        need_db = False
        if need_dx:
            dx = torch.empty((R, K), device=dy2.device, dtype=torch.float32)
            gemm16(dy16, weight16(W)[1], dx, R, K, N, N, N, K)
        if need_dw and tn:
            dW = _dw16_tn(dy16, xsave, N, K, R, dW_out)
        elif need_dw:
            sk = min(auto_splitk(N, K, R, 1), Rp // 64)
            if sk > 1:
                ws = torch.empty((sk, N * K), device=dy2.device, dtype=torch.float32)
                gemm16(dy16T, xsave, ws, N, K, Rp, Rp, Rp, K, splitk=-sk)
                dW = colsum(ws, out=None if dW_out is None else dW_out.view(-1), accumulate=dW_out is None).view(N, K)
            else:
                dW = dW_out if dW_out is not None else torch.empty((N, K), device=dy2.device, dtype=torch.float32)
                gemm16(dy16T, xsave, dW, N, K, Rp, Rp, Rp, K)
    else:
        if x16:
            raise RuntimeError("linear_bwd: bf16 activations were saved but the bf16 GEMM path is disabled now")
        if act:
            dy2 = act_bwd(dy2, act_aux, act)
        x2 = xsave
        if need_dx:
            dx = torch.empty((R, K), device=dy2.device, dtype=torch.float32)
            gemm(dy2, W, dx, R, K, N, N, K, K, False, False)
        if need_dw:
            sk = auto_splitk(N, K, R, 1)
            if sk > 1:       # slab split-K: no atomics; the slabs are summed by one column-sum launch
                ws = torch.empty((sk, N * K), device=dy2.device, dtype=torch.float32)
                gemm(dy2, x2, ws, N, K, R, N, K, K, True, False, splitk=-sk)
                dW = colsum(ws, out=None if dW_out is None else dW_out.view(-1), accumulate=dW_out is None).view(N, K)
            else:
                dW = dW_out if dW_out is not None else torch.empty((N, K), device=dy2.device, dtype=torch.float32)
                gemm(dy2, x2, dW, N, K, R, N, K, K, True, False)
    if need_db:
        db = _zeros_or(db_out, N, dy2.device)
        _call("spe_colsum", _p(dy2), _p(db), R, N, N, 1, _st())
    return dx, dW, db


def colsum(x2, out=None, accumulate=True):
    """out (zeroed, or holding a running sum) += column sums of x2; accumulate=False: out is overwritten (uninitialised memory is
    fine - what the weight-gradient slab sums into the all-reduce bucket views use, so the buckets need no zeroing)."""
    _chk(x2)
    R, C = x2.shape
    if out is None:
        out = zeros_small(C, x2.device)
    _call("spe_colsum", _p(x2), _p(out), R, C, x2.stride(0), int(bool(accumulate)), _st())
    return out


def act_bwd(dy, aux, mode):
    _chk(dy, aux)
    dx = torch.empty_like(dy)
    _call("spe_act_bwd", _p(dy), _p(aux), _p(dx), dy.numel(), mode, _st())
    return dx


# ---- LayerNorm ------------------------------------------------------------------------------
def layernorm_fwd(x2, g, b, eps, want16=False, f16=False):
    """-> (y, mean, rstd[, y16, y16lo]); want16: also the bf16 copy of y (what act16 would convert) from the same pass, and
    in the bf16s forward the low part of the split operand (else y16lo is None) - or, f16, the IEEE fp16 copy in its place (the
    consumer is a single-term fp16 product: the backbone MLP)."""
    _chk(x2, g, b)
    R, C = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty((R,), device=x2.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    y16 = torch.empty((R, C), device=x2.device, dtype=torch.bfloat16) if want16 else None
    f16 = f16 and want16 and split_fwd()
    y16lo = torch.empty((R, C), device=x2.device, dtype=torch.float16 if f16 else torch.bfloat16) if (want16 and split_fwd()) else None
    _call("spe_layernorm_fwd_h" if f16 else "spe_layernorm_fwd", _p(x2), _p(g), _p(b), _p(y), _p(mean), _p(rstd), R, C, float(eps), _p(y16), _p(y16lo), _st())
    return (y, mean, rstd, y16, y16lo) if want16 else (y, mean, rstd)


def produces16(R, C):
    """A producer of a [R, C] activation should also emit its bf16 copy: the consumer Linear takes the bf16-copy GEMM path and
    needs no transposed copy (DW_TN)."""
    return DW_TN and LINEAR16 and _PRECISION != 1 and R >= LINEAR16_MIN_ROWS and C % 8 == 0


def attach16(t, x16, x16lo=None):
    """Remember the bf16 copy (and the low part of the split operand) ON the activation tensor (what act16 looks up)."""
    t._spe16 = (t._version, x16, None, x16lo)


LN_LS_FUSE = True        # LayerScale backward of the consuming node inside the LayerNorm backward (module attribute: the tests run both settings)


def layernorm_bwd(dy2, x2, g, mean, rstd, dg_out=None, db_out=None, add=None, ls=None):
    """add [R,C]: gradient of the skip path around the normalised branch, summed into dx by the same kernel.
    ls = (y [R,C] fp32, gamma [C], db_out, dg_out): dx is the `dout` of a node out = res + gamma * y whose output feeds ONLY this norm - its
    LayerScale backward (layerscale_residual_bwd16 without rates) rides on the same pass -> (dx, dg, db, (dy16, ls_db, ls_dg))."""
    _chk(dy2, x2, g)
    R, C = x2.shape
    dx = torch.empty_like(x2)
    dg = _zeros_or(dg_out, C, x2.device)
    db = _zeros_or(db_out, C, x2.device)
    if ls is not None:
        y, lgam, ls_db_out, ls_dg_out = ls
        dy16 = torch.empty((R, C), device=x2.device, dtype=torch.bfloat16)
        ldb = _zeros_or(ls_db_out, C, x2.device)
        ldg = _zeros_or(ls_dg_out, C, x2.device)
        _call("spe_layernorm_bwd_ls", _p(dy2), _p(x2), _p(g), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), R, C, _p(add), _p(y), _p(lgam), _p(dy16),
              _p(ldb), _p(ldg), _st())
        return dx, dg, db, (dy16, ldb, ldg)
    _call("spe_layernorm_bwd", _p(dy2), _p(x2), _p(g), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), R, C, _p(add), _st())
    return dx, dg, db


def layernorm_res_fwd(x2, z2, g, b, eps, p, seed, offset):
    """norm(x + dropout(z)) -> (y, sum, mean, rstd); see spe_layernorm_res_fwd."""
    _chk(x2, z2, g, b)
    R, C = x2.shape
    y, sm = torch.empty_like(x2), torch.empty_like(x2)
    mean = torch.empty((R,), device=x2.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    _call("spe_layernorm_res_fwd", _p(x2), _p(z2), _p(g), _p(b), _p(sm), _p(y), _p(mean), _p(rstd), R, C, float(eps), float(p), seed,
          offset, _st())
    return y, sm, mean, rstd


def layernorm_res_bwd(dy2, sm, g, mean, rstd, p, seed, offset, dg_out=None, db_out=None, dy_b=None):
    """-> (ds, dz, dgamma, dbeta); dz is ds itself when p == 0.  dy_b: the gradient of a second consumer of the output (added to dy2 on load)."""
    R, C = sm.shape
    ds = torch.empty_like(sm)
    dz = torch.empty_like(sm) if p > 0 else None
    dg = _zeros_or(dg_out, C, sm.device)
    db = _zeros_or(db_out, C, sm.device)
    _call("spe_layernorm_res_bwd", _p(dy2), _p(dy_b), _p(sm), _p(g), _p(mean), _p(rstd), _p(ds), _p(dz), _p(dg), _p(db), R, C, float(p), seed,
          offset, _st())
    return ds, (dz if dz is not None else ds), dg, db


# ---- LayerScale residual --------------------------------------------------------------------
def layerscale_residual_fwd(x2, y2, gamma, sample_scale, rows_per_sample):
    _chk(x2, y2, gamma, sample_scale)
    R, C = x2.shape
    out = torch.empty_like(x2)
    _call("spe_layerscale_residual_fwd", _p(x2), _p(y2), _p(gamma), _p(sample_scale), _p(out), R, C,
             rows_per_sample, _st())
    return out


def layerscale_residual_bwd(dout2, y2, gamma, sample_scale, rows_per_sample, dg_out=None):
    _chk(dout2, y2, gamma, sample_scale)
    R, C = dout2.shape
    dy = torch.empty_like(dout2)
    dg = _zeros_or(dg_out, C, dout2.device)
    _call("spe_layerscale_residual_bwd", _p(dout2), _p(y2), _p(gamma), _p(sample_scale), _p(dy), _p(dg), R, C,
             rows_per_sample, _st())
    return dy, dg


# ---- dropout --------------------------------------------------------------------------------
def dropout(x, p, seed, offset):
    _chk(x)
    y = torch.empty_like(x)
    _call("spe_dropout", _p(x), _p(y), x.numel(), float(p), seed, offset, _st())
    return y


# ---- attention score transforms -------------------------------------------------------------
def pad4(n):
    return (n + 3) // 4 * 4


def softmax_fwd(S, mask_u8, B, H, Nq, Nk, ld, p_drop, seed, offset):
    """S [B,H,Nq,ld] -> P (in place over S), Pd (or None)."""
    Pd = torch.empty_like(S) if p_drop > 0 else None
    _call("spe_softmax_fwd", _p(S), _p(mask_u8), _p(S), _p(Pd), B, H, Nq, Nk, ld, float(p_drop), seed, offset, _st())
    return S, Pd


def softmax_bwd(dPd, P, B, H, Nq, Nk, ld, p_drop, seed, offset):
    _call("spe_softmax_bwd", _p(dPd), _p(P), _p(dPd), B, H, Nq, Nk, ld, float(p_drop), seed, offset, _st())
    return dPd


def talking_fwd(S, Wl, bl, Ww, bw, B, H, Nq, Nk, ld, p_drop, seed, offset):
    """S [B,H,Nq,ld] raw scores -> P (in place over S), Pd (new)."""
    Pd = torch.empty_like(S)
    _call("spe_talking_softmax_fwd", _p(S), _p(Wl), _p(bl), _p(Ww), _p(bw), _p(S), _p(Pd), B, H, Nq, Nk, ld,
             float(p_drop), seed, offset, _st())
    return S, Pd


def talking_bwd(dPd, P, S, Wl, Ww, B, H, Nq, Nk, ld, p_drop, seed, offset):
    """-> dS (in place over dPd), dWl, dbl, dWw, dbw."""
    nblocks = min(B * Nq, 1024)
    nw = 2 * (H * H + H)
    ws = torch.empty((nblocks, nw), device=dPd.device, dtype=torch.float32)
    _call("spe_talking_softmax_bwd", _p(dPd), _p(P), _p(S), _p(Wl), _p(Ww), _p(dPd), _p(ws), nblocks, B, H, Nq, Nk, ld,
             float(p_drop), seed, offset, _st())
    g = colsum(ws)
    hh = H * H
    return dPd, g[:hh].view(H, H), g[hh:hh + H], g[hh + H:2 * hh + H].view(H, H), g[2 * hh + H:]


# ---- misc -----------------------------------------------------------------------------------
def patchify(img, P):
    _chk(img)
    B, Cin, Hi, Wi = img.shape
    h, w = Hi // P, Wi // P
    cols = torch.empty((B * h * w, Cin * P * P), device=img.device, dtype=torch.float32)
    _call("spe_patchify", _p(img), _p(cols), B, Cin, Hi, Wi, P, _st())
    return cols


def add_rows(a, table):
    """a [.., period] + table broadcast over leading rows (a.numel() % table.numel() == 0)."""
    _chk(a, table)
    out = torch.empty_like(a)
    _call("spe_add_rows", _p(a), _p(table), _p(out), a.numel(), table.numel(), _st())
    return out


# ---- matcher / criterion --------------------------------------------------------------------
def matcher_cost(logits, boxes, tgt_ids_i32, tgt_boxes, toff_i32, total_targets, w_class, w_bbox, w_giou):
    """logits [L,B,Q,Kc], boxes [L,B,Q,4] -> cost [L, Q*total_targets] (packed per image), err flag tensor."""
    _chk(logits, boxes, tgt_boxes)
    L, B, Q, Kc = logits.shape
    cost = torch.empty((L, Q * total_targets), device=logits.device, dtype=torch.float32)
    err = torch.zeros((1,), device=logits.device, dtype=torch.int32)
    _call("spe_matcher_cost", _p(logits), _p(boxes), _p(tgt_ids_i32), _p(tgt_boxes), _p(toff_i32), total_targets,
             _p(cost), _p(err), L, B, Q, Kc, float(w_class), float(w_bbox), float(w_giou), _st())
    return cost, err


def hungarian(cost, toff_i32, L, B, Q, total_targets, err=None):
    """Device-side linear_sum_assignment of every (layer, image) block of `cost` -> (srow, gidx) int64 and lidx int32,
    each [L*total_targets] (see csrc/loss.hip: hungarian_kernel).  err: the int32 flag word of matcher_cost (bit 1 is
    raised for a problem with non-finite costs)."""
    dev = cost.device
    srow = torch.empty((L * total_targets,), device=dev, dtype=torch.int64)
    gidx = torch.empty_like(srow)
    lidx = torch.empty((L * total_targets,), device=dev, dtype=torch.int32)
    _call("spe_hungarian", _p(cost), _p(toff_i32), _p(srow), _p(gidx), _p(lidx), _p(err), L, B, Q, _st())
    return srow, gidx, lidx


def focal_loss(logits, tclass_i32, roww, alpha, gamma):
    """logits [L,R,Kc] -> loss_sum [L], grad [L,R,Kc], argmax [L,R]."""
    _chk(logits, roww)
    L, R, Kc = logits.shape
    grad = torch.empty_like(logits)
    loss = torch.zeros((L,), device=logits.device, dtype=torch.float32)
    amax = torch.empty((L, R), device=logits.device, dtype=torch.int32)
    _call("spe_focal_loss", _p(logits), _p(tclass_i32), _p(roww), _p(grad), _p(loss), _p(amax), L, R, Kc,
             float(alpha), float(gamma), _st())
    return loss, grad, amax


def box_loss(pred_boxes, srow_i64, tbox, w, lidx_i32, L):
    """-> sums [L,2], g_l1 [n,4], g_giou [n,4]."""
    _chk(pred_boxes, tbox, w)
    n = srow_i64.numel()
    sums = torch.zeros((L, 2), device=pred_boxes.device, dtype=torch.float32)
    g1 = torch.empty((n, 4), device=pred_boxes.device, dtype=torch.float32)
    g2 = torch.empty_like(g1)
    _call("spe_box_loss", _p(pred_boxes), _p(srow_i64), _p(tbox), _p(w), _p(lidx_i32), _p(sums), _p(g1), _p(g2), n, int(L), _st())
    return sums, g1, g2


def box_loss_bwd(srow_i64, lidx_i32, g1, g2, c1, c2, shape):
    dpred = torch.zeros(shape, device=g1.device, dtype=torch.float32)
    _call("spe_box_loss_bwd", _p(srow_i64), _p(lidx_i32), _p(g1), _p(g2), _p(c1), _p(c2), _p(dpred), srow_i64.numel(), _st())
    return dpred


# ---- fused talking-heads attention (bf16 mode) ----------------------------------------------
# workgroups of the statistics pass (256 CUs): it runs at 2 waves per SIMD (244 registers), i.e. 512 resident workgroups - more only adds a
# partial second round (0.188 -> 0.180 ms at cfg2 with 512 instead of 768)
STATS_NWG = 512
_STATS_NWG_SOLO = STATS_NWG


def set_cu_reserve(n):
    """Leave room for `n` foreign persistent workgroups (the channels of an RCCL ring running beside the backward: one workgroup
    each, on any CU).  The fused attention passes run ONE round of 2 workgroups per CU at 239-256 registers per lane; a CU that
    hosts a foreign wave has registers for only one of them, so the chip's capacity for these kernels is 512 - n workgroups and
    a 512-workgroup launch would need a second, nearly empty round (measured with tools/dp_proxy.py: +8 % per step for ANY
    number of foreign workgroups from 8 to 64).  With the grids cut to 512 - n the launches stay single-round.
    spe_amd.dp.GradAllReducer calls this with its channel budget when world > 1; 0 restores the solo grids."""
    global _CU_RESERVE, STATS_NWG
    _CU_RESERVE = int(n)
    STATS_NWG = max(8, (_STATS_NWG_SOLO - int(n)) & ~7) if n > 0 else _STATS_NWG_SOLO


_CU_RESERVE = 0


def get_cu_reserve():
    return _CU_RESERVE


def fused_supported(H, dh):
    """The fused attention path = statistics pass + flash forward + the two backward kernels: all of them must fit (LDS: the 8 resident tiles + 5
    stage buffers of the forward, the stage buffers + transpose tiles of the backward kernels).  H = 8 with head dim 49 .. 64 does not (no model
    of the reference has it: every CaiT variant uses head dim 48); it takes the materialised path."""
    return flash_supported(H, dh) and bwdq_supported(H, dh) and bwdk_supported(H, dh)


LOG2E = 1.4426950408889634


def attn_pack(x4, scale=1.0):
    """x4 [B,N,H,dh] view (unit last stride) -> bf16 fragment records [B,H,nt,frag_record_elems(dh)]."""
    _chk(x4)
    B, N, H, dh = x4.shape
    nt = (N + 15) // 16
    out = torch.empty((B, H, nt, frag_record_elems(dh)), device=x4.device, dtype=torch.bfloat16)
    _call("spe_attn_pack", _p(x4), x4.stride(0), x4.stride(1), x4.stride(2), B, N, H, dh, float(scale), _p(out), _st())
    return out


_PLANS = {}          # work splits are pure functions of (shape, workgroup budget): asked once per shape, not once per block and step


def fused_plan(B, N, mode=0):
    """(steps per workgroup, workgroups) the statistics pass uses (parametrises attn_merge_rows)."""
    assert mode == 0
    key = ("stats", B, N, STATS_NWG)
    r = _PLANS.get(key)
    if r is None:
        spw, nwg = ctypes.c_int(0), ctypes.c_int(0)
        lib.call("spe_talking_stats_plan", B, N, STATS_NWG, ctypes.byref(spw), ctypes.byref(nwg))
        r = _PLANS[key] = (spw.value, nwg.value)
    return r


def talking_stats(Qf, Kf, Wl, bl, ws_stats, B, H, N, dh):
    """Statistics pass: partial (max, sum) of softmax_k(Wl S + bl) per (b, head, query) -> ws_stats (B * nt * 8 * H * 32 floats); merged by
    attn_merge_rows with fused_plan(B, N)[0]."""
    _call("spe_talking_stats", _p(Qf), _p(Kf), _p(Wl), _p(bl), _p(ws_stats), B, H, N, dh, STATS_NWG, _st())


def score_blocks(B, H, N, device, dtype=torch.bfloat16):
    """Uninitialised blocked score tensor [B,H,nt,nt,64,4] (16x16 blocks; see csrc/attn_contract.hip): the backward's dS (bf16)."""
    nt = (N + 15) // 16
    return torch.empty((B, H, nt, nt, 64, 4), device=device, dtype=dtype)


def attn_pack16(x4):
    """x4: [B,N,H,dh] fp32 view (unit last stride) -> bf16 MFMA 16x16x16 A fragments [B,H,nt,ceil(dh/16),64,4]."""
    B, N, H, dh = x4.shape
    assert x4.stride(3) == 1
    nt, DT = (N + 15) // 16, (dh + 15) // 16
    out = torch.empty((B, H, nt, DT, 64, 4), device=x4.device, dtype=torch.bfloat16)
    _call("spe_attn_pack16", _p(x4), x4.stride(0), x4.stride(1), x4.stride(2), B, N, H, dh, _p(out), _st())
    return out


def frag_record_elems(dh):
    """bf16 elements of one (b, h, 16-row tile) fragment record of the score kernels: full 32-wide d-steps of 64 x 8
    plus a 16-wide tail step of 64 x 4 when dh % 32 is in 1..16 (csrc/attn_stats.hip: frag_load)."""
    rem = dh % 32
    full = dh // 32 + (1 if rem > 16 else 0)
    return full * 512 + (256 if 0 < rem <= 16 else 0)


F16 = 1000             # attn_pack_multi: kind + F16 = the same layout with fp16 elements (forward operands: q, k, v)


def attn_pack_multi(jobs):
    """jobs: list of (x4 [B,N,H,dh] fp32 view with unit last stride, scale, kind): kind 32 -> attn_pack layout (32-wide
    steps + 16-wide tail), 16 -> attn_pack16 layout, 322 -> 32-wide steps only (mha_flash); kind + F16: fp16 elements instead
    of bf16.  N and dh may differ between jobs (same B and H).  One launch; -> list of packed bf16 / fp16 tensors."""
    B, _, H, _ = jobs[0][0].shape
    n = len(jobs)
    outs = []
    for x4, scale, kind in jobs:
        Bn, N, Hn, dh = x4.shape
        assert Bn == B and Hn == H and x4.stride(3) == 1 and x4.dtype == torch.float32
        nt = (N + 15) // 16
        if kind % F16 == 32:
            shape = (B, H, nt, frag_record_elems(dh))
        elif kind % F16 == 322:
            shape = (B, H, nt, (dh + 31) // 32, 64, 8)
        else:
            shape = (B, H, nt, (dh + 15) // 16, 64, 4)
        outs.append(torch.empty(shape, device=x4.device, dtype=torch.float16 if kind >= F16 else torch.bfloat16))
    xs = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in jobs])
    strides = (ctypes.c_long * (3 * n))(*[s for j in jobs for s in (j[0].stride(0), j[0].stride(1), j[0].stride(2))])
    scales = (ctypes.c_float * n)(*[float(j[1]) for j in jobs])
    kinds = (ctypes.c_int * n)(*[{32: 0, 16: 1, 322: 2}[j[2] % F16] + (16 if j[2] >= F16 else 0) for j in jobs])
    optr = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    Ns = (ctypes.c_int * n)(*[j[0].shape[1] for j in jobs])
    dhs = (ctypes.c_int * n)(*[j[0].shape[3] for j in jobs])
    _call("spe_attn_pack_multi", n, xs, strides, scales, kinds, optr, Ns, dhs, B, H, _st())
    return outs


def mha_plan(B, H, Lq, Lk):
    nch = ctypes.c_int(0)
    lib.call("spe_mha_plan", B, H, Lq, Lk, ctypes.byref(nch))
    return nch.value


def mha_fwd(Qf, Kf, V16, mask_u8, B, H, Lq, Lk, dk, dv, nch, p_drop, seed, offset):
    """-> O [B,Lq,H*dv] fp32, LSE [B,H,Lq] (log2 domain), keepbits (dropout keep flags, None when p_drop = 0); see
    csrc/mha_flash.hip."""
    dev = Qf.device
    ntq, ntk, dvt = (Lq + 15) // 16, (Lk + 15) // 16, (dv + 15) // 16
    items = B * H * ntq * nch
    opart = torch.empty((items, dvt, 64, 4), device=dev, dtype=torch.float32)
    ml = torch.empty((items, 16, 2), device=dev, dtype=torch.float32)
    O = torch.empty((B, Lq, H * dv), device=dev, dtype=torch.float32)
    lse = torch.empty((B, H, Lq), device=dev, dtype=torch.float32)
    keep = torch.empty((B * H * ntq * ntk * 4,), device=dev, dtype=torch.int64) if p_drop > 0 else None
    _call("spe_mha_fwd", _p(Qf), _p(Kf), _p(V16), _p(mask_u8), _p(opart), _p(ml), _p(O), _p(lse), _p(keep), B, H, Lq, Lk, dk, dv,
          nch, float(p_drop), seed, offset, _st())
    return O, lse, keep


def mha_bwd(Qf, Kf, Vf, dOf, K16, Q16, dO16, mask_u8, lse, D, keep, B, H, Lq, Lk, dk, dv, nch, scale, p_drop):
    dev = Qf.device
    dq = torch.empty((B, Lq, H, dk), device=dev, dtype=torch.float32)
    # key chunks write private dq slabs that one column-sum launch adds up (atomics on the same 0.3 M addresses from 16
    # chunks cost more than the whole kernel: cross-attention backward 0.21 -> 0.14 ms)
    ws = torch.empty((nch, dq.numel()), device=dev, dtype=torch.float32) if nch > 1 else None
    dk_ = torch.empty((B, Lk, H, dk), device=dev, dtype=torch.float32)
    dv_ = torch.empty((B, Lk, H, dv), device=dev, dtype=torch.float32)
    _call("spe_mha_bwd", _p(Qf), _p(Kf), _p(Vf), _p(dOf), _p(K16), _p(Q16), _p(dO16), _p(mask_u8), _p(lse), _p(D), _p(keep), _p(dq),
          _p(ws), _p(dk_), _p(dv_), B, H, Lq, Lk, dk, dv, nch, float(scale), float(p_drop), _st())
    if ws is not None:
        colsum(ws, out=dq.view(-1), accumulate=False)          # overwrites: no zero fill in front of it
    return dq, dk_, dv_


def rowdot(x4, y4):
    """D [B,H,L] = sum_d x4 * y4 for contiguous fp32 [B,L,H,dh] tensors (the softmax backward's row term rowsum(dO . O)) in one launch."""
    _chk(x4, y4)
    B, L, H, dh = x4.shape
    D = torch.empty((B, H, L), device=x4.device, dtype=torch.float32)
    _call("spe_rowdot", _p(x4), _p(y4), _p(D), B, L, H, dh, _st())
    return D


_CONTRACT_WS = {}      # device -> (scratch floats, zeroed counters) shared by every contraction launch of the stream


def _contract_ws(device):
    ent = _CONTRACT_WS.get(device)
    if ent is None:
        ent = (torch.empty(256 * 4 * 4 * 256, device=device, dtype=torch.float32),       # 256 quarter workgroups x R*DT*256 floats (DT <= 4)
               torch.zeros(1024, device=device, dtype=torch.int32))
        _CONTRACT_WS[device] = ent
    return ent


def attn_contract(T, X16, out4, trans, alpha=1.0, out16=None, out16lo=None):
    """out4[b, row, h, :] = alpha * sum T[b,h][q,key] x[.., :]  (trans=False: rows = q, sum over keys; True: rows = keys,
    sum over q).  out4: [B,N,H,dh] fp32 view with unit last stride.  out16: optional bf16 tensor addressed with the SAME element
    strides (the bf16 copy of the result for the Linear that consumes it).  Element formats from the dtypes: bf16 T x bf16 X16;
    fp16 T x fp16 X16 (trans=False: the forward P'd V); fp16 T x bf16 X16 (trans=True: dV)."""
    ref = out4 if out4 is not None else out16          # out4 None: only the 16-bit result (out16: a view with the strides out4 would have)
    B, N, H, dh = ref.shape
    assert ref.stride(3) == 1
    tf, xf = T.dtype == torch.float16, X16.dtype == torch.float16
    fmt = {(False, False): 0, (True, True): 1, (True, False): 2}[(tf, xf)]
    ws, cnt = _contract_ws(ref.device)
    _call("spe_attn_contract", _p(T), _p(X16), _p(out4), ref.stride(0), ref.stride(1), ref.stride(2), B, H, N, dh,
          int(trans), fmt, float(alpha), _p(ws), _p(cnt), ws.numel(), _p(out16), _p(out16lo), _st())
    return ref


# ---- flash-style talking-heads attention (csrc/attn_flash.hip): no N x N tensor in HBM -------------------------------------
# workgroups: 8-wave workgroups (two waves per SIMD), one per CU
FLASH_NWG = 256
FLASH_SLOTS = 8


def flash_supported(H, dh):
    """The flash kernels keep 8 resident tiles and 5 stage buffers (+ 3 KB of row constants) in LDS."""
    return H in (4, 8) and dh <= 64 and 13 * H * ((dh + 15) // 16) * 512 + 3072 <= 160 * 1024


def flash_plan(B, N):
    """(steps per workgroup, workgroups, major tile groups per image, padded rows of the row-constant arrays)."""
    key = ("flash", B, N, FLASH_NWG)
    r = _PLANS.get(key)
    if r is None:
        spw, nwg, nmaj, npad = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        lib.call("spe_talking_flash_plan", B, N, FLASH_NWG, ctypes.byref(spw), ctypes.byref(nwg), ctypes.byref(nmaj), ctypes.byref(npad))
        r = _PLANS[key] = (spw.value, nwg.value, nmaj.value, npad.value)
    return r


_FLASH_WS = {}         # device -> partial-result workspace shared by every flash launch of the stream (all blocks reuse one)


def _flash_ws(device, floats):
    ent = _FLASH_WS.get(device)
    if ent is None or ent.numel() < floats:
        ent = torch.empty((floats,), device=device, dtype=torch.float32)
        _FLASH_WS[device] = ent
    return ent


def talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, c0, B, H, N, dh, p_drop, seed, offset, want16=False, want16lo=False, want_bits=False):
    """-> O [B,N,H*dh] fp32 (+ bf16 copy, + its low part; want_bits with dropout: + the keep flags [B,nt,nt,64] int32 of every tile
    for the backward's q-major passes - appended as a 4th result)."""
    dev = Qf.device
    nmaj = flash_plan(B, N)[2]
    ws = _flash_ws(dev, B * nmaj * FLASH_SLOTS * 128 * H * 16 * ((dh + 15) // 16))
    C = H * dh
    O = torch.empty((B, N, C), device=dev, dtype=torch.float32)
    O16 = torch.empty((B * N, C), device=dev, dtype=torch.bfloat16) if want16 else None
    O16lo = torch.empty((B * N, C), device=dev, dtype=torch.bfloat16) if (want16 and want16lo) else None
    nt = (N + 15) // 16
    bits = torch.empty((B, nt, nt, 64), device=dev, dtype=torch.int32) if (want_bits and p_drop > 0) else None
    _call("spe_talking_flash_fwd", _p(Qf), _p(Kf), _p(V16), _p(Wl), _p(Ww), _p(bw), _p(c0), c0.shape[1], _p(ws), _p(O), _p(O16), _p(O16lo),
          _p(bits), B, H, N, dh, FLASH_NWG, float(p_drop), seed, offset, _st())
    if want_bits:
        return O, O16, O16lo, bits
    return O, O16, O16lo


# ---- the two backward kernels (csrc/attn_flash_bwd.hip): 4-wave workgroups, one wave per SIMD, one workgroup per CU
def bwdq_supported(H, dh):
    DT = (dh + 15) // 16
    return H in (4, 8) and dh <= 64 and 7 * H * DT * 512 + 512 + 4 * 3 * 4 * H * 144 <= 160 * 1024


def bwdq_plan(B, N):
    """(steps per workgroup, workgroups, major (4 q-tile) groups per image) of the q-major backward passes."""
    budget = max(8, FLASH_NWG - _CU_RESERVE)
    key = ("bwdq", B, N, budget)
    r = _PLANS.get(key)
    if r is None:
        spw, nwg, nmaj = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        lib.call("spe_talking_bwdq_plan", B, N, budget, ctypes.byref(spw), ctypes.byref(nwg), ctypes.byref(nmaj))
        if len(_PLANS) > 256:
            _PLANS.clear()
        r = _PLANS[key] = (spw.value, nwg.value, nmaj.value)
    return r


def bwdk_supported(H, dh):
    DT = (dh + 15) // 16
    return H in (4, 8) and dh <= 64 and 7 * H * DT * 512 + 2048 + 2 * 4 * 16 * H * 4 + 512 + 4 * 3 * 4 * H * 144 <= 160 * 1024


def talking_bwdk_pass1(Qf, dOf, dO16, Kf, Vf, Wl, Ww, bw, c0, keepbits, dv4, dv16, B, H, N, dh, p_drop):
    """KEY-major backward pass 1 + dV in one launch (csrc/attn_flash_bwd.hip, talking_bwdk_kernel) -> (Drows [B, Np, H], ws_w [4 nwg, 2 (H H + H)]
    with its dWw / dbw half filled); dv = P'd^T dO lands in dv4 (fp32 view [B,N,H,dh]) and / or dv16 (bf16 view, same strides)."""
    ref = dv4 if dv4 is not None else dv16
    assert ref.stride(3) == 1 and (dv4 is None or dv16 is None or dv4.stride() == dv16.stride())
    dev = Qf.device
    _, nwg, nmaj = bwdq_plan(B, N)
    Np = c0.shape[1]
    ws_d = torch.empty((B * nmaj, Np, H), device=dev, dtype=torch.float32)
    ws_v = _flash_ws(dev, B * nmaj * FLASH_SLOTS * 4 * H * 256 * ((dh + 15) // 16))
    ws_w = torch.empty((4 * nwg, 2 * (H * H + H)), device=dev, dtype=torch.float32)
    Drows = torch.empty((B, Np, H), device=dev, dtype=torch.float32)
    _call("spe_talking_bwdk_pass1", _p(Qf), _p(dOf), _p(dO16), _p(Kf), _p(Vf), _p(Wl), _p(Ww), _p(bw), _p(c0), Np, _p(ws_d), _p(ws_v), _p(ws_w),
          _p(Drows), _p(dv4), _p(dv16), ref.stride(0), ref.stride(1), ref.stride(2), _p(keepbits), B, H, N, dh, max(8, FLASH_NWG - _CU_RESERVE),
          float(p_drop), _st())
    return Drows, ws_w


def talking_bwdq_pass2(Qf, dOf, Kf, Vf, K16, Wl, Ww, c0, Drows, ws_w, dS, dq4, dq16, scale, keepbits, B, H, N, dh, p_drop):
    """dS (bf16 blocks) and dq = scale dS k (dq4 fp32 view [B,N,H,dh] and / or dq16 bf16 view with the same strides); fills the dWl / dbl half
    of ws_w."""
    ref = dq4 if dq4 is not None else dq16
    assert ref.stride(3) == 1 and (dq4 is None or dq16 is None or dq4.stride() == dq16.stride())
    nmaj = bwdq_plan(B, N)[2]
    ws_q = _flash_ws(ref.device, B * nmaj * FLASH_SLOTS * 4 * H * 256 * ((dh + 15) // 16))
    _call("spe_talking_bwdq_pass2", _p(Qf), _p(dOf), _p(Kf), _p(Vf), _p(K16), _p(Wl), _p(Ww), _p(c0), _p(Drows), c0.shape[1], _p(ws_q), _p(ws_w),
          _p(dS), _p(dq4), _p(dq16), ref.stride(0), ref.stride(1), ref.stride(2), float(scale), _p(keepbits), B, H, N, dh,
          max(8, FLASH_NWG - _CU_RESERVE), float(p_drop), _st())


def attn_merge_rows(ws_stats, bl, B, H, N, spw):
    """Merge of the statistics pass -> (M, IL [B,H,N], c0 [B,Np,H]: the row constants of the flash forward and the backward kernels)."""
    Np = flash_plan(B, N)[3]
    M = torch.empty((B, H, N), device=ws_stats.device, dtype=torch.float32)
    IL = torch.empty_like(M)
    c0 = torch.empty((B, Np, H), device=ws_stats.device, dtype=torch.float32)
    _call("spe_attn_merge_rows", _p(ws_stats), _p(M), _p(IL), _p(bl), _p(c0), Np, B, H, N, spw, _st())
    return M, IL, c0


def jitter_pick(box, scale, ratio):
    """box [M,4] cxcywh, scale [M,ncand,4] uniform factors -> [M, ratio, 4]: the first ratio - 1 candidates with IoU > 0.7, original last."""
    _chk(box, scale)
    M, ncand = scale.shape[0], scale.shape[1]
    out = torch.empty((M, ratio, 4), device=box.device, dtype=torch.float32)
    _call("spe_jitter_pick", _p(box), _p(scale), _p(out), M, ncand, int(ratio), _st())
    return out


def talking_wgrad_reduce(ws_w, H, params):
    """Column sums of the weight-gradient partials [nwg, 2*(H*H+H)] as (dWl [H,H], dbl [H], dWw [H,H], dbw [H]), written
    into the parameters' all-reduce bucket views when those are still unclaimed this step (params = Wl, bl, Ww, bw; a None entry: a fresh tensor).
    The kernel STORES its sums (no accumulation)."""
    shapes = ((H, H), (H,), (H, H), (H,))
    outs = []
    for prm, shp in zip(params, shapes):
        buf = grad_buffer(prm) if prm is not None else None
        outs.append(buf.view(shp) if buf is not None else torch.empty(shp, device=ws_w.device, dtype=torch.float32))
    _call("spe_talking_wgrad_reduce", _p(ws_w), ws_w.shape[0], H, _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]), _st())
    return outs


def gemm_bf16a(A16, B, C, M, N, K, lda, ldb, ldc, transA, transB, batch0, batch1, sA, sB, sC, alpha=1.0):
    """GEMM whose A operand is a bf16 tensor (element strides)."""
    _call("spe_gemm_ex", _p(A16), 1, _p(B), _p(C), None, None, M, N, K, lda, ldb, ldc, int(transA), int(transB), batch0, batch1,
          sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], float(alpha), 0, 1, 0, _st())
    return C


def bicubic(src, gh, gw, h, w, backward=False):
    """Token-major bicubic resize of a [gh*gw, C] grid to [h*w, C] (or its adjoint when backward)."""
    _chk(src)
    C = src.shape[-1]
    if backward:
        out = torch.zeros((gh * gw, C), device=src.device, dtype=torch.float32)
    else:
        out = torch.empty((h * w, C), device=src.device, dtype=torch.float32)
    _call("spe_bicubic", _p(src), _p(out), gh, gw, h, w, C, int(backward), _st())
    return out


# ---- optimiser (flat buffers) ----------------------------------------------------------------
def sqnorm_partials(g_flat, partials):
    _call("spe_sqnorm_partials", _p(g_flat), g_flat.numel(), _p(partials), partials.numel(), _st())


def adamw_flat(p, g, m, v, seg_end_i64, seg_lr, seg_wd, beta1, beta2, eps, bias_c1, bias_c2, partials, max_norm, write_grad,
               grad_scale=1.0):
    _call("spe_adamw_flat", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(seg_end_i64), _p(seg_lr), _p(seg_wd), seg_end_i64.numel(),
          float(beta1), float(beta2), float(eps), float(bias_c1), float(bias_c2), _p(partials), partials.numel(),
          float(max_norm), int(bool(write_grad)), float(grad_scale), _st())


# ---- inference post-processing -----------------------------------------------------------------
def nms_sorted(boxes, labels, iou_threshold, counts=None):
    """boxes [I,n,4] fp32 xyxy, labels [I,n] int64, per image ordered by (label asc, score desc) -> keep mask [I,n] bool."""
    _chk(boxes)
    I, n, _ = boxes.shape
    assert labels.dtype == torch.int64 and labels.is_contiguous()
    keep = torch.empty((I, n), device=boxes.device, dtype=torch.uint8)
    _call("spe_nms_sorted", _p(boxes), _p(labels), _p(counts), _p(keep), I, n, float(iou_threshold), _st())
    return keep.bool()


# ---- CAM -> pseudo boxes -------------------------------------------------------------------------
def cam_prepare(maps, rows, cols, cam_thr):
    """maps [M,h,w] fp32 (device) -> thresholded uint8 images [M,rows,cols] (device); see csrc/cambox.hip."""
    _chk(maps)
    M, h, w = maps.shape
    out = torch.empty((M, rows, cols), device=maps.device, dtype=torch.uint8)
    mm = torch.empty((2 * M,), device=maps.device, dtype=torch.float32)
    _call("spe_cam_prepare", _p(maps), M, h, w, rows, cols, float(cam_thr), _p(mm), _p(out), _st())
    return out


def cam_contour_boxes(img_u8_host, area_ratio, max_boxes=256):
    """One thresholded uint8 image on the HOST ([rows, cols], contiguous) -> int32 [n,4] boxes [x, y, x+w, y+h]."""
    assert img_u8_host.device.type == "cpu" and img_u8_host.dtype == torch.uint8 and img_u8_host.is_contiguous()
    rows, cols = img_u8_host.shape
    boxes = torch.empty((max_boxes, 4), dtype=torch.int32)
    n = ctypes.c_int(0)
    lib.call("spe_cam_contour_boxes", ctypes.c_void_p(img_u8_host.data_ptr()), rows, cols, float(area_ratio),
             ctypes.c_void_p(boxes.data_ptr()), max_boxes, ctypes.byref(n))
    return boxes[:n.value].clone()


def pos_sine(mask_bool, dim_t, npf, scale, eps, normalize):
    """mask [B,h,w] bool (True = padded) -> [B,h,w,2*npf] fp32 sine position features (csrc/misc.hip)."""
    B, h, w = mask_bool.shape
    m8 = mask_bool.to(torch.uint8).contiguous()
    out = torch.empty((B, h, w, 2 * npf), device=mask_bool.device, dtype=torch.float32)
    _call("spe_pos_sine", _p(m8), _p(dim_t), _p(out), B, h, w, npf, float(scale), float(eps), int(bool(normalize)), _st())
    return out
