"""Autograd operators of the SPE hot path.  Every forward AND backward is a sequence of
libspe_hip.so kernel launches (spe_amd/kernels.py); torch only owns memory and the graph.

Layout convention inside spe_amd: activations are batch-first [B, L, D] (the reference's
transformer uses [L, B, D]; only the public outputs follow the reference's layout).
"""
import os

import torch
from torch.autograd import Function

from . import kernels as K

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


# ---------------------------------------------------------------------------------------------
class _Linear(Function):
    """y = act(x W^T + b)  (nn.Linear [+ F.relu / nn.GELU]); reference transformer.py:21-33,
    cait.py:376,390, timm Mlp."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, W, b, act):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        Wc = W if W.is_contiguous() else W.contiguous()
        y, pre, xsave = K.linear_fwd(x2, Wc, b, act, want_pre=(act == ACT_GELU), save_for_dw=ctx.needs_input_grad[1], src=x)
        aux = pre if act == ACT_GELU else (y if act == ACT_RELU else None)
        ctx.act = act
        ctx.has_bias = b is not None
        ctx.params = (W, b)                 # leaves: looked up in backward for their gradient buckets
        ctx.save_for_backward(xsave, Wc, aux)
        return y.view(*shp[:-1], W.shape[0])

    @staticmethod
    @K.backward_scope
    def backward(ctx, dy):
        x2, W, aux = ctx.saved_tensors
        dy2 = dy.reshape(-1, W.shape[0])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        Wp, bp = ctx.params
        need_dw, need_db = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx, dW, db = K.linear_bwd(dy2, x2, W, ctx.needs_input_grad[0], need_dw, need_db,
                                  dW_out=K.grad_buffer(Wp) if need_dw and Wp.is_contiguous() else None,
                                  db_out=K.grad_buffer(bp) if need_db else None, act=ctx.act, act_aux=aux)
        if dx is not None:
            dx = dx.view(*dy.shape[:-1], W.shape[1])
        return dx, dW, db, None


def linear(x, W, b=None, act=ACT_NONE):
    return _Linear.apply(x, W, b, act)


class _MultiLinear(Function):
    """(x W_0^T + b_0, x W_1^T + b_1, ...) for several nn.Linear of equal shape applied to ONE input - the query-independent
    memory-side projections of the conditional cross attention of ALL decoder layers (reference models/transformer.py:
    389-396: ca_kcontent_proj / ca_v_proj of `memory`, ca_kpos_proj of `pos`, evaluated per layer and per decoder pass there) as one
    [R, K] x [K, n*N] GEMM whose column blocks are the n results (views of one buffer); the input gradient is one GEMM over the
    stacked weights (instead of n GEMMs and n - 1 full-size adds), the weight gradients are n row-major TN products on column
    blocks of the stacked bf16 dy, the bias gradients ride on the n conversion passes."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, *wb):
        n = len(wb) // 2
        Ws, bs = wb[:n], wb[n:]
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        R, Kd = x2.shape
        N = Ws[0].shape[0]
        sp = K.split_fwd()                    # bf16s forward: (hi, lo) operand pairs
        cat = K.weightcat16(Ws, bs, lo=sp)
        Wc, WcT, bc = cat[:3]
        x16, _, x16lo = K.act16(x2, False, x, want_lo=sp)
        y = torch.empty((R, n * N), device=x.device, dtype=torch.float32)
        K.gemm16(x16, Wc, y, R, n * N, Kd, Kd, Kd, n * N, bias=bc, Alo=x16lo, Blo=cat[3] if sp else None)
        ctx.params = (Ws, bs)
        ctx.save_for_backward(x16, WcT)
        ctx.dims = (n, N, Kd, R)
        y = y.view(*shp[:-1], n * N)
        return tuple(y[..., i * N:(i + 1) * N] for i in range(n))

    @staticmethod
    @K.backward_scope
    def backward(ctx, *dys):
        x16, WcT = ctx.saved_tensors
        n, N, Kd, R = ctx.dims
        Ws, bs = ctx.params
        dev = x16.device
        dy16 = torch.empty((R, n * N), device=dev, dtype=torch.bfloat16)
        dWs, dbs = [], []
        for i in range(n):
            blk = dy16[:, i * N:(i + 1) * N]
            if dys[i] is None:
                blk.zero_()
                dWs.append(None); dbs.append(None)
                continue
            d2 = dys[i].reshape(R, N)
            if not d2.is_contiguous():
                d2 = d2.contiguous()
            db = K._zeros_or(K.grad_buffer(bs[i]), N, dev)
            K.cvt_bf16(d2, True, False, colsum_out=db, out=blk, ldo=n * N)
            dbs.append(db)
            dWs.append(K._dw16_tn(blk, x16, N, Kd, R, K.grad_buffer(Ws[i]), lda=n * N))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((R, Kd), device=dev, dtype=torch.float32)
            K.gemm16(dy16, WcT, dx, R, Kd, n * N, n * N, n * N, Kd)
            dx = dx.view(*dys[0].shape[:-1], Kd) if dys[0] is not None else dx
        return (dx, *dWs, *dbs)


def multi_linear_ok(x, Ws, bs):
    R = x.numel() // x.shape[-1]
    N, Kd = Ws[0].shape
    return (K.DW_TN and K._lin16_ok(R, len(Ws) * N, Kd) and all(W.shape == Ws[0].shape and W.is_contiguous() for W in Ws)
            and all(b is not None for b in bs) and N % 8 == 0)


def multi_linear(x, Ws, bs):
    """-> tuple of n tensors [.., N] (column blocks of one buffer); element i is linear(x, Ws[i], bs[i])."""
    return _MultiLinear.apply(x, *Ws, *bs)


class _GroupLinear(Function):
    """(x W_0^T + b_0, ..., x W_{n-1}^T + b_{n-1}) for n nn.Linear of one shape on ONE input of a few hundred rows - the query side of a
    conditional-DETR decoder layer (reference models/transformer.py:368-372: sa_qcontent_proj / sa_kcontent_proj / sa_v_proj of tgt;
    369-371, 399: sa_qpos_proj / sa_kpos_proj of every layer and the first layer's ca_qpos_proj of query_pos) - as ONE launch each way
    (csrc/linear_small.hip group entries) instead of n forward launches, n backward launches and n - 1 gradient-accumulation adds;
    same arithmetic per output as ops.linear on that path.  Optional addends (n_add = n, entries may be None): output i + adds[i] from the same launch
    (transformer.py:373-374: q = q_content + q_pos, k = k_content + k_pos); an addend's gradient is the output's gradient itself."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, n_add, *rest):
        adds, wb = rest[:n_add], rest[n_add:]
        n = len(wb) // 2
        Ws, bs = wb[:n], wb[n:]
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        N = Ws[0].shape[0]
        a2 = None
        if n_add:
            a2 = [None if a is None else (lambda u: u if u.is_contiguous() else u.contiguous())(a.reshape(-1, N)) for a in adds]
        ys, x16 = K.linear_group_fwd(x2, Ws, bs, src=x, adds=a2)
        ctx.set_materialize_grads(False)          # an output outside the loss arrives as None, not as a zero tensor
        ctx.params = (Ws, bs)
        ctx.add_shapes = tuple(None if a is None else a.shape for a in adds)
        ctx.save_for_backward(x16)
        return tuple(y.view(*shp[:-1], N) for y in ys)

    @staticmethod
    @K.backward_scope
    def backward(ctx, *dys):
        (x16,) = ctx.saved_tensors
        Ws, bs = ctx.params
        n, N = len(Ws), Ws[0].shape[0]
        d2 = []
        for d in dys:
            if d is None:
                d2.append(None)
                continue
            d = d.reshape(-1, N)
            d2.append(d if d.is_contiguous() else d.contiguous())
        # bucket views are claimed only for the outputs that received a gradient (a claimed view must be written)
        dx, dWs, dbs = K.linear_group_bwd(d2, x16, Ws, ctx.needs_input_grad[0],
                                          [K.grad_buffer(W) if d is not None else None for W, d in zip(Ws, d2)],
                                          [K.grad_buffer(b) if d is not None else None for b, d in zip(bs, d2)])
        if dx is not None:
            ref = next(d for d in dys if d is not None)
            dx = dx.view(*ref.shape[:-1], Ws[0].shape[1])
        dadds = tuple(None if (shp is None or d is None) else d.reshape(shp) for shp, d in zip(ctx.add_shapes, dys))
        return (dx, None, *dadds, *dWs, *[None if b is None else b.view_as(p) for b, p in zip(dbs, bs)])


def group_linear_ok(x, mods):
    """mods: nn.Linear-like modules (weight, bias) of one shape; True when the one-launch group form applies to x."""
    if not x.is_cuda or len(mods) < 2:
        return False
    R = x.numel() // x.shape[-1]
    return K.linear_group_ok(R, [m.weight for m in mods], [m.bias for m in mods])


def group_linear(x, mods, adds=None):
    """adds (optional): one entry per module - a tensor of the output's shape that is added to that output in the same launch, or None."""
    adds = tuple(adds) if adds is not None else ()
    assert len(adds) in (0, len(mods))
    return _GroupLinear.apply(x, len(adds), *adds, *[m.weight for m in mods], *[m.bias for m in mods])


# ---------------------------------------------------------------------------------------------
class _LayerNorm(Function):
    @staticmethod
    @K.forward_scope
    def forward(ctx, x, g, b, eps):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        y, mean, rstd = K.layernorm_fwd(x2, g, b, eps)
        ctx.params = (g, b)
        ctx.save_for_backward(x2, g, mean, rstd)
        return y.view(x.shape)

    @staticmethod
    @K.backward_scope
    def backward(ctx, dy):
        x2, g, mean, rstd = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        gp, bp = ctx.params
        dx, dg, db = K.layernorm_bwd(dy2, x2, g, mean, rstd, dg_out=K.grad_buffer(gp), db_out=K.grad_buffer(bp))
        return dx.view(dy.shape), dg.view_as(gp), db.view_as(bp), None


def layer_norm(x, g, b, eps):
    return _LayerNorm.apply(x, g, b, eps)


class _ResDropLayerNorm(Function):
    """norm(x + dropout(z)) - a post-norm residual site of the DETR encoder / decoder layers (reference
    models/transformer.py:279-287, 384-386, 420-421, 426-427) as one kernel each way instead of dropout, add and LayerNorm."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, z, g, b, eps, p):
        x2 = x.reshape(-1, x.shape[-1])
        z2 = z.reshape(-1, z.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        z2 = z2 if z2.is_contiguous() else z2.contiguous()
        seed, off = K.next_rng() if p > 0 else (0, 0)
        y, sm, mean, rstd = K.layernorm_res_fwd(x2, z2, g, b, eps, p, seed, off)
        ctx.params = (g, b)
        ctx.drop = (p, seed, off)
        ctx.save_for_backward(sm, g, mean, rstd)
        return y.view(x.shape)

    @staticmethod
    @K.backward_scope
    def backward(ctx, dy):
        sm, g, mean, rstd = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        gp, bp = ctx.params
        p, seed, off = ctx.drop
        ds, dz, dg, db = K.layernorm_res_bwd(dy2, sm, g, mean, rstd, p, seed, off, dg_out=K.grad_buffer(gp), db_out=K.grad_buffer(bp))
        return ds.view(dy.shape), dz.view(dy.shape), dg.view_as(gp), db.view_as(bp), None, None


def res_drop_layer_norm(x, z, g, b, eps, p, training):
    return _ResDropLayerNorm.apply(x, z, g, b, eps, p if training else 0.0)


class _ResDropLayerNorm2(Function):
    """(y, y) with y = norm(x + dropout(z)) for a post-norm layer whose output has TWO consumers - a Linear and the next residual site (reference
    models/transformer.py:384-427: tgt = norm1(...); q = ca_qcontent_proj(tgt); ...; tgt = norm2(tgt + ...)): the two gradients of y meet in THIS node and
    are added while the LayerNorm backward loads its rows (spe_layernorm_res_bwd dy2), instead of by an autograd accumulation launch."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, z, g, b, eps, p):
        x2 = x.reshape(-1, x.shape[-1])
        z2 = z.reshape(-1, z.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        z2 = z2 if z2.is_contiguous() else z2.contiguous()
        seed, off = K.next_rng() if p > 0 else (0, 0)
        y, sm, mean, rstd = K.layernorm_res_fwd(x2, z2, g, b, eps, p, seed, off)
        ctx.params = (g, b)
        ctx.drop = (p, seed, off)
        ctx.save_for_backward(sm, g, mean, rstd)
        yv = y.view(x.shape)
        return yv, yv.view_as(yv)

    @staticmethod
    @K.backward_scope
    def backward(ctx, dy, dy_b):
        sm, g, mean, rstd = ctx.saved_tensors
        if dy is None:
            dy, dy_b = dy_b, None
        if dy is None:
            return None, None, None, None, None, None
        c2 = lambda t: None if t is None else (lambda u: u if u.is_contiguous() else u.contiguous())(t.reshape(-1, t.shape[-1]))
        gp, bp = ctx.params
        p, seed, off = ctx.drop
        ds, dz, dg, db = K.layernorm_res_bwd(c2(dy), sm, g, mean, rstd, p, seed, off, dg_out=K.grad_buffer(gp), db_out=K.grad_buffer(bp), dy_b=c2(dy_b))
        return ds.view(dy.shape), dz.view(dy.shape), dg.view_as(gp), db.view_as(bp), None, None


def res_drop_layer_norm2(x, z, g, b, eps, p, training):
    """-> (y, y_skip): use the first for the layer's next Linear, the second as the skip operand of the next residual site."""
    return _ResDropLayerNorm2.apply(x, z, g, b, eps, p if training else 0.0)


class _LayerNormSkip(Function):
    """(LN(x), x) for a pre-norm residual branch x + f(LN(x)) (reference models/cait.py:404-405): the second output is x
    itself, to be used as the residual operand, so that the two gradients of x - through the normalisation and over the
    skip path - meet in THIS node and are summed inside the LayerNorm backward kernel instead of by an extra autograd add."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, g, b, eps, f16=False):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if K.produces16(*x2.shape):          # the Linear behind this norm runs on bf16 copies: emit its operand here
            # (f16: the consumer is a single-term fp16 product - the MLP of precision mode bf16s -: the fp16 copy instead of the low part)
            y, mean, rstd, y16, y16lo = K.layernorm_fwd(x2, g, b, eps, want16=True, f16=f16)
            yv = y.view(x.shape)
            K.attach16(yv, y16, y16lo)
        else:
            y, mean, rstd = K.layernorm_fwd(x2, g, b, eps)
            yv = y.view(x.shape)
        ctx.params = (g, b)
        # the node that produced x, when it is a LayerScale-residual Linear / MLP node whose output feeds ONLY this norm (its caller said so: `single`):
        # this norm's backward then also takes that node's LayerScale backward - gamma * dx as bf16, the bias and gamma column sums - from the dx it is
        # writing anyway (kernels.layernorm_bwd ls=), instead of that node re-reading dx in a launch of its own.
        # CONTRACT of `single`: the producer's backward runs in the same backward pass as this norm's (always true for loss.backward() of the
        # model; NOT for torch.autograd.grad(..., inputs=<the block output>) that stops between the two nodes - its bias / gamma bucket views
        # would then hold this norm's partial sums only).  A producer that cannot consume what was taken raises (kernels.linear_res_bwd / mlp_gelu_bwd)
        prod = x.grad_fn if (K.LN_LS_FUSE and x.requires_grad) else None
        ctx.prod = prod if (prod is not None and getattr(prod, "ls_single", False) and x2.shape[1] <= 512) else None
        ctx.save_for_backward(x2, g, mean, rstd)
        return yv, x.view_as(x)

    @staticmethod
    @K.backward_scope
    def backward(ctx, dy, dskip):
        x2, g, mean, rstd = ctx.saved_tensors
        gp, bp = ctx.params
        if dy is None:                       # only the skip path was used
            return dskip, None, None, None, None
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        add = None
        if dskip is not None:
            add = dskip.reshape(-1, dskip.shape[-1])
            if not add.is_contiguous():
                add = add.contiguous()
        prod = ctx.prod
        if prod is not None:
            # (y, gamma) of the producing node and the bucket views of its bias / gamma gradients; the node finds the results in prod.ls_done
            yls, gls, bpar, gpar = _ls_args(prod)
            dx, dg, db, done = K.layernorm_bwd(dy2, x2, g, mean, rstd, dg_out=K.grad_buffer(gp), db_out=K.grad_buffer(bp), add=add,
                                               ls=(yls, gls, K.grad_buffer(bpar), K.grad_buffer(gpar)))
            dxv = dx.view(dy.shape)
            # (address AND version: the engine may add a second consumer's gradient into this very buffer in place - same address, version + 1)
            prod.ls_done = (done, dx.data_ptr(), dxv._version)
            return dxv, dg.view_as(gp), db.view_as(bp), None, None
        dx, dg, db = K.layernorm_bwd(dy2, x2, g, mean, rstd, dg_out=K.grad_buffer(gp), db_out=K.grad_buffer(bp), add=add)
        return dx.view(dy.shape), dg.view_as(gp), db.view_as(bp), None, None


def layer_norm_skip(x, g, b, eps, f16=False):
    """-> (LN(x), x): use the second result as the residual operand of the branch (see _LayerNormSkip)."""
    return _LayerNormSkip.apply(x, g, b, eps, f16)


# ---------------------------------------------------------------------------------------------
class _LayerScaleResidual(Function):
    """out = x + s_b * gamma * y   (cait.py:413-416; s_b = DropPath keep-scale per sample)."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, y, gamma, sample_scale):
        B = x.shape[0]
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y2 = y.reshape(-1, y.shape[-1]).contiguous()
        rps = x2.shape[0] // B
        out = K.layerscale_residual_fwd(x2, y2, gamma, sample_scale, rps)
        ctx.rps = rps
        ctx.gamma = gamma
        ctx.save_for_backward(y2, gamma, sample_scale)
        return out.view(x.shape)

    @staticmethod
    @K.backward_scope
    def backward(ctx, dout):
        y2, gamma, ss = ctx.saved_tensors
        d2 = dout.reshape(-1, dout.shape[-1]).contiguous()
        dy, dg = K.layerscale_residual_bwd(d2, y2, gamma, ss, ctx.rps, dg_out=K.grad_buffer(ctx.gamma))
        return dout, dy.view(dout.shape), dg.view_as(ctx.gamma), None


def layerscale_residual(x, y, gamma, sample_scale=None):
    return _LayerScaleResidual.apply(x, y, gamma, sample_scale)


def drop_path_scale(B, p, training, device):
    """Per-sample keep/(1-p) factors of timm DropPath (reference models/layers/drop.py:140-168)."""
    if p <= 0.0 or not training:
        return None
    keep = 1.0 - p
    return (torch.rand(B, device=device) < keep).to(torch.float32) / keep


# ---------------------------------------------------------------------------------------------
class _Dropout(Function):
    @staticmethod
    @K.forward_scope
    def forward(ctx, x, p):
        seed, off = K.next_rng()
        ctx.p, ctx.seed, ctx.off = p, seed, off
        return K.dropout(x.contiguous(), p, seed, off)

    @staticmethod
    @K.backward_scope
    def backward(ctx, dy):
        return K.dropout(dy.contiguous(), ctx.p, ctx.seed, ctx.off), None


def dropout(x, p, training):
    if p <= 0.0 or not training:
        return x
    return _Dropout.apply(x, p)


# ---------------------------------------------------------------------------------------------
class _TalkingHeadsAttention(Function):
    """CaiT talking-heads self-attention core (reference models/cait.py:377-389): qkv [B,N,3C] ->
    out [B,N,C].  Materialised-score implementation: QK^T GEMM -> fused head-mix/softmax/head-mix/
    dropout row kernel -> PV GEMM; the saved tensors are P and Pd ([B,H,N,N] each)."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, qkv, Wl, bl, Ww, bw, H, scale, p_drop):
        B, N, C3 = qkv.shape
        C = C3 // 3
        dh = C // H
        qkv = qkv.contiguous()
        v5 = qkv.view(B, N, 3, H, dh)
        q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
        ld = K.pad4(N)
        S = torch.empty((B, H, N, ld), device=qkv.device, dtype=torch.float32)
        sq = (N * C3, dh)
        sS = (H * N * ld, N * ld)
        K.gemm(q, k, S, N, N, dh, C3, C3, ld, False, True, batch0=B, batch1=H, sA=sq, sB=sq, sC=sS, alpha=scale)
        seed, off = K.next_rng() if p_drop > 0 else (0, 0)
        P, Pd = K.talking_fwd(S, Wl, bl, Ww, bw, B, H, N, N, ld, p_drop, seed, off)
        O = torch.empty((B, N, C), device=qkv.device, dtype=torch.float32)
        K.gemm(Pd, v, O, N, dh, N, ld, C3, C, False, False, batch0=B, batch1=H, sA=sS, sB=sq, sC=(N * C, dh))
        ctx.meta = (B, N, C, H, dh, ld, scale, p_drop, seed, off)
        ctx.save_for_backward(qkv, P, Pd, Wl, Ww)
        return O

    @staticmethod
    @K.backward_scope
    def backward(ctx, dO):
        qkv, P, Pd, Wl, Ww = ctx.saved_tensors
        B, N, C, H, dh, ld, scale, p_drop, seed, off = ctx.meta
        C3 = 3 * C
        dO = dO.contiguous()
        v5 = qkv.view(B, N, 3, H, dh)
        q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
        dqkv = torch.empty_like(qkv)
        d5 = dqkv.view(B, N, 3, H, dh)
        dq, dk, dv = d5[:, :, 0], d5[:, :, 1], d5[:, :, 2]
        sq, sS, sO = (N * C3, dh), (H * N * ld, N * ld), (N * C, dh)
        # dPd = dO V^T ; dV = Pd^T dO
        dPd = torch.empty((B, H, N, ld), device=qkv.device, dtype=torch.float32)
        K.gemm(dO, v, dPd, N, N, dh, C, C3, ld, False, True, batch0=B, batch1=H, sA=sO, sB=sq, sC=sS)
        K.gemm(Pd, dO, dv, N, dh, N, ld, C, C3, True, False, batch0=B, batch1=H, sA=sS, sB=sO, sC=sq)
        # raw scores again (cheaper than keeping a third [B,H,N,N] tensor alive per block)
        S = torch.empty((B, H, N, ld), device=qkv.device, dtype=torch.float32)
        K.gemm(q, k, S, N, N, dh, C3, C3, ld, False, True, batch0=B, batch1=H, sA=sq, sB=sq, sC=sS, alpha=scale)
        dS, dWl, dbl, dWw, dbw = K.talking_bwd(dPd, P, S, Wl, Ww, B, H, N, N, ld, p_drop, seed, off)
        del S
        # dQ = scale * dS K ; dK = scale * dS^T Q
        K.gemm(dS, k, dq, N, dh, N, ld, C3, C3, False, False, batch0=B, batch1=H, sA=sS, sB=sq, sC=sq, alpha=scale)
        K.gemm(dS, q, dk, N, dh, N, ld, C3, C3, True, False, batch0=B, batch1=H, sA=sS, sB=sq, sC=sq, alpha=scale)
        return dqkv, dWl, dbl, dWw, dbw, None, None, None


class _TalkingHeadsAttentionFused(Function):
    """Same operator on the fused kernels: no N x N tensor in HBM in the forward, only the bf16 dS in the backward.  ONE composition:
    forward : pack q * scale * log2 e, k, v (fp16 fragment records) -> statistics pass (csrc/attn_stats.hip) -> merge (row constants c0) -> flash
              forward (csrc/attn_flash.hip: S, S', P, P', dropout, O += P'd V in registers; with dropout it stores the 1-bit keep flags)
    backward: pack dO -> key-major kernel (D, dWw, dbw, dV) -> query-major kernel (dS blocks, dWl, dbl, dQ) -> weight-gradient reduce -> dK
              contraction (csrc/attn_flash_bwd.hip, csrc/attn_contract.hip)
    Saved for backward: the packed q / k fragments (fp16 for the score recompute, bf16 for the gradient contractions), the bf16 v fragments, the
    row constants and - with dropout - the keep flags; not qkv itself."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, qkv, Wl, bl, Ww, bw, H, scale, p_drop):
        train = any(ctx.needs_input_grad)
        O, saved = _TalkingHeadsAttentionFused._fwd(ctx, qkv, Wl, bl, Ww, bw, H, scale, p_drop, train)
        if train:
            ctx.save_for_backward(*saved)
        return O

    @staticmethod
    def _fwd(ctx, qkv, Wl, bl, Ww, bw, H, scale, p_drop, train):
        """-> (O, tensors the backward needs); also used by _QkvTalkingAttention (the qkv Linear inside the same node)."""
        B, N, C3 = qkv.shape
        C = C3 // 3
        dh = C // H
        qkv = qkv.contiguous()
        v5 = qkv.view(B, N, 3, H, dh)
        q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
        nt = (N + 15) // 16
        spw0, _ = K.fused_plan(B, N)
        # the fused kernels work in the log2 domain: scale * log2(e) is folded into the Q fragments
        # forward operands in fp16 (O(1) values: 3 more mantissa bits than bf16 at the same size and MFMA rate)
        # ... and, when a backward will follow, its bf16 fragments of q / k / v from the same read of qkv (one launch; the fp32
        # qkv - 38 MB per block at cfg2 - is then neither re-read nor kept)
        jobs = [(q, scale * K.LOG2E, 32 + K.F16), (k, 1.0, 32 + K.F16), (v, 1.0, 16 + K.F16)]
        if train:
            jobs += [(v, 1.0, 32), (k, 1.0, 16), (q, 1.0, 16)]
        packed = K.attn_pack_multi(jobs)
        Qf, Kf, V16 = packed[:3]
        Wl, bl, Ww, bw = Wl.contiguous(), bl.contiguous(), Ww.contiguous(), bw.contiguous()
        ws_stats = torch.empty((B * nt * 8 * H * 32,), device=qkv.device, dtype=torch.float32)
        seed, off = K.next_rng() if p_drop > 0 else (0, 0)
        K.talking_stats(Qf, Kf, Wl, bl, ws_stats, B, H, N, dh)
        want16 = K.produces16(B * N, C)
        _, _, c0 = K.attn_merge_rows(ws_stats, bl, B, H, N, spw0)
        # P' goes from the head mix straight into the P' V products: nothing N x N is stored or saved - the backward recomputes it
        O, O16, O16lo, bits = K.talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, c0, B, H, N, dh, p_drop, seed, off, want16, K.split_fwd(),
                                                  want_bits=True)
        if O16 is not None:
            K.attach16(O, O16, O16lo)        # the output projection's operand, written by the merge's epilogue
        ctx.meta = (B, N, C, H, dh, nt, scale, p_drop)
        ctx.wparams = (Wl, bl, Ww, bw)      # leaves: looked up in backward for their gradient buckets
        # with dropout the flash forward leaves the keep flags of every tile behind (1 bit per element): the backward kernels load them
        ctx.has_bits = train and bits is not None
        saved = (packed[3], packed[4], packed[5], Qf, Kf, c0, Wl, Ww, bw) + ((bits,) if ctx.has_bits else ()) if train else ()
        return O, saved

    @staticmethod
    @K.backward_scope
    def backward(ctx, dO):
        return (*_TalkingHeadsAttentionFused._bwd(ctx, ctx.saved_tensors, dO, False), None, None, None)

    @staticmethod
    def _bwd(ctx, saved, dO, out16):
        """-> (dqkv, dWl, dbl, dWw, dbw).  out16: dqkv comes back as the bf16 [B, N, 3C] operand of the qkv Linear's backward GEMMs
        (written by the contraction / merge epilogues); no fp32 copy exists then."""
        Vf, K16, Q16, Qf, Kf, c0, Wl, Ww, bw = saved[:9]
        kbits = saved[9] if ctx.has_bits else None
        B, N, C, H, dh, nt, scale, p_drop = ctx.meta
        dO = dO.contiguous()
        dqkv = torch.empty((B, N, 3 * C), device=dO.device, dtype=torch.bfloat16 if out16 else torch.float32)
        d5 = dqkv.view(B, N, 3, H, dh)
        dq, dk, dv = d5[:, :, 0], d5[:, :, 1], d5[:, :, 2]
        f32 = (lambda t: None) if out16 else (lambda t: t)          # the fp32 destination of a gradient (None: 16-bit only)
        b16 = (lambda t: t) if out16 else (lambda t: None)
        dO4 = dO.view(B, N, H, dh)
        dOf, dO16 = K.attn_pack_multi([(dO4, 1.0, 32), (dO4, 1.0, 16)])
        dS = K.score_blocks(B, H, N, dO.device)
        # KEY-major kernel: S, S', P recomputed once for D, dWw, dbw AND dV ; then the QUERY-major kernel: dS blocks, dWl, dbl and dQ in registers
        Drows, ws_w = K.talking_bwdk_pass1(Qf, dOf, dO16, Kf, Vf, Wl, Ww, bw, c0, kbits, f32(dv), b16(dv), B, H, N, dh, p_drop)
        K.talking_bwdq_pass2(Qf, dOf, Kf, Vf, K16, Wl, Ww, c0, Drows, ws_w, dS, f32(dq), b16(dq), scale, kbits, B, H, N, dh, p_drop)
        dWl, dbl, dWw, dbw = K.talking_wgrad_reduce(ws_w, H, ctx.wparams)
        # dK[key,d] = scale * sum_q dS[q,key] Q[q,d]: the one streaming read of dS
        K.attn_contract(dS, Q16, f32(dk), True, alpha=scale, out16=b16(dk))
        return dqkv, dWl, dbl, dWw, dbw


class _QkvTalkingAttention(Function):
    """qkv Linear + talking-heads attention of a backbone block (reference models/cait.py:376-389) as ONE autograd node: the gradient
    w.r.t. qkv exists only as the bf16 operand of the Linear's backward GEMMs, written by the epilogues of the dQ / dK contractions and
    the dV merge - no fp32 [B, N, 3C] gradient (38 MB per block at cfg2) is written, re-read and converted.  Same arithmetic as
    ops.linear followed by _TalkingHeadsAttentionFused otherwise (the Linear's backward rounds that gradient to bf16 as well)."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, Wq, bq, Wl, bl, Ww, bw, H, scale, p_drop):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        y, _, xsave = K.linear_fwd(x2, Wq, bq, 0, want_pre=False, save_for_dw=True, src=x)
        O, saved = _TalkingHeadsAttentionFused._fwd(ctx, y.view(*shp[:-1], Wq.shape[0]), Wl, bl, Ww, bw, H, scale, p_drop, True)
        ctx.qparams = (Wq, bq)
        ctx.save_for_backward(xsave, *saved)
        return O

    @staticmethod
    @K.backward_scope
    def backward(ctx, dO):
        xsave, *saved = ctx.saved_tensors
        Wq, bq = ctx.qparams
        d16, dWl, dbl, dWw, dbw = _TalkingHeadsAttentionFused._bwd(ctx, saved, dO, True)
        N3, Kd = Wq.shape
        R = d16.numel() // N3
        d16 = d16.view(R, N3)
        dev = d16.device
        gb = K.grad_buffer(bq)
        db = gb.view(-1) if gb is not None else torch.empty((N3,), device=dev, dtype=torch.float32)
        K.colsum_bf16_blocks(d16, N3, [db])                            # overwrites, fixed summation order
        dW = K._dw16_tn(d16, xsave, N3, Kd, R, K.grad_buffer(Wq)) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((R, Kd), device=dev, dtype=torch.float32)
            K.gemm16(d16, K.weight16(Wq)[1], dx, R, Kd, N3, N3, N3, Kd)
            dx = dx.view(*dO.shape[:-1], Kd)
        return dx, dW, db.view_as(bq), dWl, dbl, dWw, dbw, None, None, None


QKV_FUSED = True        # module attribute, not an environment knob


def qkv_talking_attention_ok(x, Wq, bq, num_heads):
    """The one-node form needs the bf16-operand Linear path with row-major saves, a bias, the fused attention kernels and gradients."""
    C = x.shape[-1]
    R = x.numel() // C
    return (QKV_FUSED and x.is_cuda and bq is not None and Wq.is_contiguous() and torch.is_grad_enabled()
            and (x.requires_grad or Wq.requires_grad) and Wq.requires_grad and bq.requires_grad
            and K.get_precision() != "bf16x3" and K.DW_TN and K._lin16_ok(R, Wq.shape[0], C) and not K._lin_small_ok(R, Wq.shape[0], C)
            and K.fused_supported(num_heads, C // num_heads))


def qkv_talking_attention(x, Wq, bq, Wl, bl, Ww, bw, num_heads, scale, p_drop=0.0):
    return _QkvTalkingAttention.apply(x, Wq, bq, Wl, bl, Ww, bw, num_heads, scale, p_drop)


class _MlpGelu(Function):
    """fc2(gelu(fc1(x))) of the backbone block (reference models/cait.py:405-412: timm Mlp with drop = 0) as ONE autograd
    node on the bf16-copy GEMMs: the activation and the gradient w.r.t. the pre-activation never exist in fp32 - the
    producing GEMM epilogues write the bf16 (and transposed bf16) operands of the following GEMMs and the bias gradient."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, W1, b1, W2, b2):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        train = any(ctx.needs_input_grad)
        y, saved = K.mlp_gelu_fwd(x2, W1, b1, W2, b2, save=train, src=x)
        ctx.params = (W1, b1, W2, b2)
        if train:
            ctx.save_for_backward(*saved, W1, W2)
        return y.view(*shp[:-1], W2.shape[0])

    @staticmethod
    @K.backward_scope
    def backward(ctx, dy):
        x16T, pre, h16T, W1, W2 = ctx.saved_tensors
        dy2 = dy.reshape(-1, W2.shape[0])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        bufs = tuple(K.grad_buffer(p) for p in ctx.params)
        dx, dW1, db1, dW2, db2 = K.mlp_gelu_bwd(dy2, (x16T, pre, h16T), W1, W2, ctx.needs_input_grad[0], bufs)
        return (dx.view(*dy.shape[:-1], W1.shape[1]) if dx is not None else None), dW1, db1, dW2, db2


FUSE_LINEAR_RES = True


class _LinearRes(Function):
    """xres + gamma * linear(x) - the attention half of the backbone block after the attention itself (reference
    models/cait.py:390 `proj` inside x + gamma_1 * attn(norm1(x)), :404-405; drop_path = proj_drop = 0) as one node: the
    LayerScale residual rides on the projection GEMM's epilogue and the backward emits gamma * dout directly as the bf16
    operands of the projection's gradient GEMMs."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, W, b, xres, gamma, sscale=None, p_drop=0.0, single=False):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        r2 = xres.reshape(-1, xres.shape[-1])
        if not r2.is_contiguous():
            r2 = r2.contiguous()
        train = any(ctx.needs_input_grad)
        # training rates of the block (cait.py:391 proj_drop, :404 drop_path) ride on the epilogue: dropout mask of spe_dropout's stream,
        # DropPath keep scale per sample
        ctx.drop = (float(p_drop), *K.next_rng()) if p_drop > 0 else None
        # `single`: the caller promises that the output feeds one LayerNorm-skip node and nothing else - that norm's backward may then take this
        # node's LayerScale backward along (ops._LayerNormSkip); only without rates, with fp32 saves and bias / gamma gradients wanted
        ctx.ls_single = bool(single and train and p_drop <= 0 and sscale is None and not K.MLP_PRE_F16 and K.DW_TN and W.requires_grad
                             and b.requires_grad and gamma.requires_grad)
        ctx.ls_done = None
        ctx.rps = r2.shape[0] // xres.shape[0]
        out, saved = K.linear_res_fwd(x2, W, b, r2, gamma, save=train, src=x, drop=ctx.drop, sscale=sscale, rps=ctx.rps)
        ctx.params = (W, b, gamma)
        ctx.has_ss = sscale is not None
        if train:
            ctx.save_for_backward(*saved, W, gamma, *((sscale,) if sscale is not None else ()))
        return out.view(xres.shape)

    @staticmethod
    @K.backward_scope
    def backward(ctx, dout):
        x16T, y, W, gamma = ctx.saved_tensors[:4]
        ss = ctx.saved_tensors[4] if ctx.has_ss else None
        d2 = dout.reshape(-1, W.shape[0])
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        pre = _ls_taken(ctx, dout if dout.is_contiguous() else d2)
        bufs = tuple(K.grad_buffer(p) for p in ctx.params)
        dx, dW, db, dg = K.linear_res_bwd(d2, (x16T, y), W, gamma, ctx.needs_input_grad[0], bufs, drop=ctx.drop, sscale=ss, rps=ctx.rps, pre=pre)
        return (dx.view(*dout.shape[:-1], W.shape[1]) if dx is not None else None), dW, db, dout, dg.view_as(gamma), None, None, None

def _ls_args(prod):
    """(branch output y, gamma, bias parameter, gamma parameter) of a LayerScale-residual node (the ctx of a _LinearRes / _MlpGeluRes: an instance of
    the generated backward class, which carries `_forward_cls`) for the LayerNorm backward that takes its LayerScale backward along."""
    sv = prod.saved_tensors
    if prod._forward_cls is _LinearRes:
        return sv[1], sv[3], prod.params[1], prod.params[2]
    return sv[3], sv[6], prod.params[3], prod.params[4]           # _MlpGeluRes


def _ls_taken(ctx, d2):
    """The (dy16, db, dgamma) a LayerNorm backward left for this node (see _LayerNormSkip), or None.  The sums are already in the gradients: the
    incoming gradient must be the very tensor that backward wrote - anything else means the single-consumer promise was broken."""
    done = getattr(ctx, "ls_done", None)
    if done is None:
        return None
    ctx.ls_done = None
    pre, ptr, ver = done
    if d2.data_ptr() != ptr or d2._version != ver:
        raise RuntimeError("spe_amd.ops: a node was marked single=True but its output had more than one consumer (the LayerNorm backward already "
                           "accumulated its bias / LayerScale gradients from a partial gradient)")
    return pre


FUSE_DROP = True      # dropout / DropPath inside the fused residual nodes (module attribute: tests/test_round4_gpu.py runs both settings)


def linear_residual(x, W, b, xres, gamma, sample_scale=None, p_drop=0.0, single=False):
    """xres + s * gamma * dropout(linear(x)): one fused node on the bf16-copy GEMM path (the training rates ride on its epilogue), else
    linear, dropout and layerscale_residual.  single: the result feeds ONE LayerNorm-skip node and nothing else (see _LinearRes)."""
    R = x.numel() // x.shape[-1]
    N, Kd = W.shape
    if (FUSE_LINEAR_RES and (FUSE_DROP or (sample_scale is None and p_drop <= 0)) and b is not None and W.is_contiguous()
            and gamma.is_contiguous() and N % 4 == 0 and N <= 1024 and K._lin16_ok(R, N, Kd)
            and (sample_scale is None or (sample_scale.is_contiguous() and R % xres.shape[0] == 0))):
        return _LinearRes.apply(x, W, b, xres, gamma, sample_scale, p_drop, single)
    return layerscale_residual(xres, dropout(linear(x, W, b), p_drop, p_drop > 0), gamma, sample_scale)


class _MlpGeluRes(Function):
    """xres + gamma * fc2(gelu(fc1(x))) - the MLP half of the backbone block after its LayerNorm (reference
    models/cait.py:405-416, drop_path = 0) as one node: _MlpGelu plus the LayerScale residual in the fc2 epilogue, and in
    the backward the branch gradient gamma * dout emitted directly as the bf16 operands of the fc2 gradient GEMMs."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, W1, b1, W2, b2, xres, gamma, sscale=None, p_drop=0.0, single=False):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        r2 = xres.reshape(-1, xres.shape[-1])
        if not r2.is_contiguous():
            r2 = r2.contiguous()
        train = any(ctx.needs_input_grad)
        # timm Mlp: fc1 -> GELU -> drop -> fc2 -> drop, then DropPath and the LayerScale residual (cait.py:405-416): the two dropout
        # sites draw their streams in that order, like the unfused composition
        ctx.drop1 = (float(p_drop), *K.next_rng()) if p_drop > 0 else None
        ctx.drop2 = (float(p_drop), *K.next_rng()) if p_drop > 0 else None
        ctx.ls_single = bool(single and train and p_drop <= 0 and sscale is None and not K.MLP_PRE_F16 and K.DW_TN and W2.requires_grad
                             and b2.requires_grad and gamma.requires_grad)          # see _LinearRes
        ctx.ls_done = None
        ctx.rps = r2.shape[0] // xres.shape[0]
        out, saved = K.mlp_gelu_fwd(x2, W1, b1, W2, b2, res=r2, gamma=gamma, save=train, src=x, drop1=ctx.drop1, drop2=ctx.drop2,
                                    sscale=sscale, rps=ctx.rps)
        ctx.params = (W1, b1, W2, b2, gamma)
        ctx.has_ss = sscale is not None
        if train:
            ctx.save_for_backward(*saved, W1, W2, gamma, *((sscale,) if sscale is not None else ()))
        return out.view(xres.shape)

    @staticmethod
    @K.backward_scope
    def backward(ctx, dout):
        x16T, pre, h16T, y, W1, W2, gamma = ctx.saved_tensors[:7]
        ss = ctx.saved_tensors[7] if ctx.has_ss else None
        d2 = dout.reshape(-1, W2.shape[0])
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        W1p, b1p, W2p, b2p, gp = ctx.params
        taken = _ls_taken(ctx, dout if dout.is_contiguous() else d2)
        bufs = tuple(K.grad_buffer(p) for p in (W1p, b1p, W2p, b2p))
        dx, dW1, db1, dW2, db2, dg = K.mlp_gelu_bwd(d2, (x16T, pre, h16T, y), W1, W2, ctx.needs_input_grad[0], bufs,
                                                     gamma=gamma, dg_out=K.grad_buffer(gp), drop1=ctx.drop1, drop2=ctx.drop2,
                                                     sscale=ss, rps=ctx.rps, ls_pre=taken)
        return ((dx.view(*dout.shape[:-1], W1.shape[1]) if dx is not None else None), dW1, db1, dW2, db2, dout,
                dg.view_as(gamma), None, None, None)


def mlp_gelu_residual(x, W1, b1, W2, b2, xres, gamma, sample_scale=None, p_drop=0.0, single=False):
    """xres + s * gamma * drop(fc2(drop(gelu(fc1(x))))).  One fused node when the MLP takes the bf16-copy path (dropout and the per-sample
    DropPath scale ride on the GEMM epilogues); otherwise the composition of the single operators (same arithmetic, same masks)."""
    R = x.numel() // x.shape[-1]
    if ((FUSE_DROP or (sample_scale is None and p_drop <= 0)) and b1 is not None and b2 is not None and W1.is_contiguous() and W2.is_contiguous()
            and gamma.is_contiguous() and W2.shape[0] % 4 == 0 and W1.shape[0] % 4 == 0 and W2.shape[0] <= 1024
            and K.mlp16_ok(R, W1.shape[1], W1.shape[0], W2.shape[0])
            and (sample_scale is None or (sample_scale.is_contiguous() and R % xres.shape[0] == 0))):
        return _MlpGeluRes.apply(x, W1, b1, W2, b2, xres, gamma, sample_scale, p_drop, single)
    if p_drop > 0:
        h = dropout(linear(x, W1, b1, ACT_GELU), p_drop, True)
        return layerscale_residual(xres, dropout(linear(h, W2, b2), p_drop, True), gamma, sample_scale)
    return layerscale_residual(xres, mlp_gelu(x, W1, b1, W2, b2), gamma, sample_scale)


def mlp_gelu(x, W1, b1, W2, b2):
    """Fused MLP when both Linears take the bf16-copy path (benchmark precision, >= LINEAR16_MIN_ROWS rows, contiguous
    weights with biases); otherwise the two Linear nodes (same arithmetic, fp32 intermediates)."""
    R = x.numel() // x.shape[-1]
    if (b1 is not None and b2 is not None and W1.is_contiguous() and W2.is_contiguous()
            and K.mlp16_ok(R, W1.shape[1], W1.shape[0], W2.shape[0])):
        return _MlpGelu.apply(x, W1, b1, W2, b2)
    return linear(linear(x, W1, b1, ACT_GELU), W2, b2)


def talking_heads_attention(qkv, Wl, bl, Ww, bw, num_heads, scale, p_drop=0.0, fused=None):
    """fused=None: use the fused kernels in the bf16 / bf16s modes when the head geometry is supported (their forward runs
    on fp16 operands in both modes, csrc/attn_fused.hip); the 3-term (bf16x3) parity mode keeps the fp32 materialised path."""
    dh = qkv.shape[-1] // (3 * num_heads)
    if fused is None:
        fused = K.get_precision() != "bf16x3" and K.fused_supported(num_heads, dh)
    fn = _TalkingHeadsAttentionFused if fused else _TalkingHeadsAttention
    return fn.apply(qkv, Wl, bl, Ww, bw, num_heads, scale, p_drop)


# ---------------------------------------------------------------------------------------------
def _bh(t):
    """(batch stride, head stride, row stride) of a [B,L,H,d] view with unit last stride."""
    assert t.stride(3) == 1, "last dim must be contiguous"
    return (t.stride(0), t.stride(2)), t.stride(1)


class _Attention(Function):
    """softmax(scale * q k^T + key_padding_mask) [dropout] v for q [B,Lq,H,dk], k [B,Lk,H,dk],
    v [B,Lk,H,dv] -> [B,Lq,H*dv] (and optionally the softmax map [B,H,Lq,Lk], no grad).
    Reference: models/attention.py:277-383 (decoder MHA; q/k head dim != v head dim),
    nn.MultiheadAttention core (encoder, transformer.py:275-277), Multi_Class_Attention
    (cait.py:120-131, map saved at :130)."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, q, k, v, mask_u8, scale, p_drop, need_map):
        B, Lq, H, dk = q.shape
        Lk, dv = k.shape[1], v.shape[3]
        sQ, ldq = _bh(q)
        sK, ldk = _bh(k)
        sV, ldv = _bh(v)
        ld = K.pad4(Lk)
        S = torch.empty((B, H, Lq, ld), device=q.device, dtype=torch.float32)
        sS = (H * Lq * ld, Lq * ld)
        K.gemm(q, k, S, Lq, Lk, dk, ldq, ldk, ld, False, True, batch0=B, batch1=H, sA=sQ, sB=sK, sC=sS, alpha=scale)
        seed, off = K.next_rng() if p_drop > 0 else (0, 0)
        P, Pd = K.softmax_fwd(S, mask_u8, B, H, Lq, Lk, ld, p_drop, seed, off)
        O = torch.empty((B, Lq, H * dv), device=q.device, dtype=torch.float32)
        K.gemm_splitk_into(Pd if Pd is not None else P, v, O, Lq, dv, Lk, ld, ldv, H * dv, False, False, B, H, sS, sV,
                           (Lq * H * dv, dv))
        ctx.meta = (B, Lq, Lk, H, dk, dv, ld, scale, p_drop, seed, off)
        ctx.save_for_backward(q, k, v, P, Pd)
        if need_map:
            pmap = P[..., :Lk]
            ctx.mark_non_differentiable(pmap)
            return O, pmap
        return O, None

    @staticmethod
    @K.backward_scope
    def backward(ctx, dO, _dmap):
        q, k, v, P, Pd = ctx.saved_tensors
        B, Lq, Lk, H, dk, dv, ld, scale, p_drop, seed, off = ctx.meta
        dO = dO.contiguous()
        sQ, ldq = _bh(q)
        sK, ldk = _bh(k)
        sV, ldv = _bh(v)
        sS = (H * Lq * ld, Lq * ld)
        sO = (Lq * H * dv, dv)
        dq = torch.empty((B, Lq, H, dk), device=q.device, dtype=torch.float32)
        dk_ = torch.empty((B, Lk, H, dk), device=q.device, dtype=torch.float32)
        dv_ = torch.empty((B, Lk, H, dv), device=q.device, dtype=torch.float32)
        dP = torch.empty((B, H, Lq, ld), device=q.device, dtype=torch.float32)
        K.gemm(dO, v, dP, Lq, Lk, dv, H * dv, ldv, ld, False, True, batch0=B, batch1=H, sA=sO, sB=sV, sC=sS)
        K.gemm(Pd if Pd is not None else P, dO, dv_, Lk, dv, Lq, ld, H * dv, H * dv, True, False, batch0=B, batch1=H,
               sA=sS, sB=sO, sC=(Lk * H * dv, dv))
        dS = K.softmax_bwd(dP, P, B, H, Lq, Lk, ld, p_drop, seed, off)
        K.gemm_splitk_into(dS, k, dq, Lq, dk, Lk, ld, ldk, H * dk, False, False, B, H, sS, sK, (Lq * H * dk, dk), alpha=scale)
        K.gemm(dS, q, dk_, Lk, dk, Lq, ld, ldq, H * dk, True, False, batch0=B, batch1=H, sA=sS, sB=sQ,
               sC=(Lk * H * dk, dk), alpha=scale)
        return dq, dk_, dv_, None, None, None, None


class _AttentionFlash(Function):
    """Same operator without the [B,H,Lq,Lk] tensors (csrc/mha_flash.hip): online-softmax forward, two-kernel backward.
    Used in bf16 mode when no attention map is requested and the head dims fit (q/k <= 96, v <= 64)."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, q, k, v, mask_u8, scale, p_drop):
        B, Lq, H, dk = q.shape
        Lk, dv = k.shape[1], v.shape[3]
        sc = scale * K.LOG2E
        need_bwd = any(ctx.needs_input_grad[:3])
        # forward operands (q, k for the scores - also recomputed by the backward - and v for P.V) in fp16; the 16-wide q / k and the
        # 32-wide v fragments meet gradients in the backward and stay bf16
        jobs = [(q, sc, 322 + K.F16), (k, 1.0, 322 + K.F16), (v, 1.0, 16 + K.F16)]
        if need_bwd:
            jobs += [(q, sc, 16), (k, 1.0, 16), (v, 1.0, 322)]
        packs = K.attn_pack_multi(jobs)
        Qf, Kf, V16 = packs[:3]
        nch = K.mha_plan(B, H, Lq, Lk)
        seed, off = K.next_rng() if p_drop > 0 else (0, 0)
        O, lse, keep = K.mha_fwd(Qf, Kf, V16, mask_u8, B, H, Lq, Lk, dk, dv, nch, p_drop, seed, off)
        ctx.meta = (B, Lq, Lk, H, dk, dv, nch, scale, p_drop)
        if need_bwd:
            ctx.save_for_backward(Qf, Kf, packs[3], packs[4], packs[5], O, lse, mask_u8, keep)
        return O

    @staticmethod
    @K.backward_scope
    def backward(ctx, dO):
        Qf, Kf, Q16, K16, Vf, O, lse, mask_u8, keep = ctx.saved_tensors
        B, Lq, Lk, H, dk, dv, nch, scale, p_drop = ctx.meta
        dO = dO.contiguous()
        dO4 = dO.view(B, Lq, H, dv)
        D = K.rowdot(dO4, O.contiguous().view(B, Lq, H, dv))                             # [B,H,Lq] = rowsum(dO . O)
        dOf, dO16 = K.attn_pack_multi([(dO4, 1.0, 322), (dO4, 1.0, 16)])
        dq, dk_, dv_ = K.mha_bwd(Qf, Kf, Vf, dOf, K16, Q16, dO16, mask_u8, lse, D, keep, B, H, Lq, Lk, dk, dv, nch, scale, p_drop)
        return dq, dk_, dv_, None, None, None


class MemoryKV:
    """What the memory-side node hands to the cross-attention nodes of the decoder layers (not a tensor: fragments + the shared
    gradient buffers the layers' backward passes fill)."""

    def __init__(self):
        self.Kf = self.V16 = self.K16 = self.Vf = None
        self.dYm = self.dYp = None          # bf16 [B*S, 2*L*d] / [B*S, L*d]: allocated by the first layer's backward of a step
        self.written = set()
        self.dims = None                    # (L, B, S, H, dh)
        self.ztok = None


class _MemorySideKV(Function):
    """ca_kcontent_proj / ca_v_proj (of `memory`) and ca_kpos_proj (of `pos`) of ALL decoder layers and the per-head key layout
    [k_content | k_pos] (reference models/transformer.py:389-419) -> the operand fragments of the flash MHA kernels, without an fp32
    key / value tensor: two stacked GEMMs on IEEE fp16 single-term operands with fp16 outputs (north_star's "decoder cross-attention
    GEMM": spe_gemm_bf16nt act bits 8 + 9) and ONE fragment launch (spe_kv_frags).  Outputs: one scalar token per layer - the
    autograd edges to the layers' cross-attention nodes, which write their key / value gradients straight into the bf16 dY operands
    of this node's backward GEMMs (MemoryKV.dYm / dYp) and return a zero for the token."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, holder, memory, pos, H, *wb):
        L = len(wb) // 6
        Wm, Wp, bm, bp = wb[:2 * L], wb[2 * L:3 * L], wb[3 * L:5 * L], wb[5 * L:]
        B, S, d = memory.shape
        dh = d // H
        m2, p2 = memory.reshape(B * S, d), pos.reshape(B * S, d)
        m2 = m2 if m2.is_contiguous() else m2.contiguous()
        p2 = p2 if p2.is_contiguous() else p2.contiguous()
        # the bf16 backward fragments follow whether ANY backward can happen (holder.want_bwd: grad mode at the call site): with the memory-side
        # projections frozen and the query side trainable, the cross-attention nodes still need K16 / Vf for dq
        train = any(ctx.needs_input_grad) or bool(getattr(holder, "want_bwd", False))
        Wmh, bmc = K.weightcat_f16(Wm, bm)
        Wph, bpc = K.weightcat_f16(Wp, bp)
        ym = torch.empty((B * S, 2 * L * d), device=memory.device, dtype=torch.float16)
        yp = torch.empty((B * S, L * d), device=memory.device, dtype=torch.float16)
        K.gemm16(K.cvt_f16(m2), Wmh, ym, B * S, 2 * L * d, d, d, d, 2 * L * d, bias=bmc, act=0x300)
        K.gemm16(K.cvt_f16(p2), Wph, yp, B * S, L * d, d, d, d, L * d, bias=bpc, act=0x300)
        holder.Kf, holder.V16, holder.K16, holder.Vf = K.kv_frags(ym, yp, L, B, S, H, dh, train)
        holder.dims = (L, B, S, H, dh)
        holder.ztok = torch.zeros((1,), device=memory.device, dtype=torch.float32)
        ctx.holder = holder
        ctx.params = (Wm, Wp, bm, bp)
        if any(ctx.needs_input_grad):
            # the backward runs on single bf16 operands like every other Linear backward: bf16 copies of the inputs (shared with the
            # other consumers of `memory` / `pos`) and the transposed bf16 weight stack for the input gradient
            m16, _, _ = K.act16(m2, False, memory)
            p16, _, _ = K.act16(p2, False, pos)
            ctx.save_for_backward(m16, p16, K.weightcat16(Wm, bm)[1])
        toks = tuple(torch.zeros((1,), device=memory.device, dtype=torch.float32) for _ in range(L))
        return toks

    @staticmethod
    @K.backward_scope
    def backward(ctx, *dtoks):
        m16, p16, WmT = ctx.saved_tensors
        h = ctx.holder
        L, B, S, H, dh = h.dims
        d, R = H * dh, B * S
        Wm, Wp, bm, bp = ctx.params
        dev = m16.device
        if h.dYm is None:
            h.dYm = torch.zeros((R, 2 * L * d), device=dev, dtype=torch.bfloat16)
            h.dYp = torch.zeros((R, L * d), device=dev, dtype=torch.bfloat16)
            h.written = set(range(L))
        for l in range(L):           # a layer whose cross attention got no gradient this step: zeros
            if l not in h.written:
                h.dYm[:, 2 * l * d:(2 * l + 2) * d].zero_()
                h.dYp[:, l * d:(l + 1) * d].zero_()
        dbm = [K.grad_buffer(b) for b in bm]
        dbm = [g.view(-1) if g is not None else torch.empty((d,), device=dev, dtype=torch.float32) for g in dbm]
        dbp = [K.grad_buffer(b) for b in bp]
        dbp = [g.view(-1) if g is not None else torch.empty((d,), device=dev, dtype=torch.float32) for g in dbp]
        K.colsum_bf16_blocks(h.dYm, d, dbm)
        K.colsum_bf16_blocks(h.dYp, d, dbp)
        dWm = [K._dw16_tn(h.dYm[:, i * d:(i + 1) * d], m16, d, d, R, K.grad_buffer(Wm[i]), lda=2 * L * d) for i in range(2 * L)]
        dWp = [K._dw16_tn(h.dYp[:, i * d:(i + 1) * d], p16, d, d, R, K.grad_buffer(Wp[i]), lda=L * d) for i in range(L)]
        dmem = None
        if ctx.needs_input_grad[1]:
            dmem = torch.empty((R, d), device=dev, dtype=torch.float32)
            K.gemm16(h.dYm, WmT, dmem, R, d, 2 * L * d, 2 * L * d, 2 * L * d, d)
            dmem = dmem.view(B, S, d)
        dpos = None
        if ctx.needs_input_grad[2]:
            dpos = torch.empty((R, d), device=dev, dtype=torch.float32)
            K.gemm16(h.dYp, K.weightcat16(Wp, bp)[1], dpos, R, d, L * d, L * d, L * d, d)
            dpos = dpos.view(B, S, d)
        h.dYm = h.dYp = None
        h.written = set()
        return (None, dmem, dpos, None, *dWm, *dWp, *[g.view_as(b) for g, b in zip(dbm, bm)], *[g.view_as(b) for g, b in zip(dbp, bp)])


class _CrossAttentionKV(Function):
    """softmax(scale q k^T + key_padding_mask) [dropout] v for one decoder layer (reference models/attention.py:277-383) on the
    fragments of _MemorySideKV: q [B,Lq,H,2 dh] fp32, keys / values by (holder, layer).  The key / value gradients do not travel
    through autograd as fp32 tensors: they are scattered as bf16 into the memory-side node's dY buffers (spe_kv_grad_scatter)."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, q, tok, holder, layer, mask_u8, scale, p_drop):
        B, Lq, H, dk = q.shape
        L, _, S, _, dh = holder.dims
        sc = scale * K.LOG2E
        need_bwd = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        jobs = [(q, sc, 322 + K.F16)] + ([(q, sc, 16)] if need_bwd else [])
        packs = K.attn_pack_multi(jobs)
        nch = K.mha_plan(B, H, Lq, S)
        seed, off = K.next_rng() if p_drop > 0 else (0, 0)
        O, lse, keep = K.mha_fwd(packs[0], holder.Kf[layer], holder.V16[layer], mask_u8, B, H, Lq, S, dk, dh, nch, p_drop, seed, off)
        ctx.meta = (B, Lq, S, H, dk, dh, nch, scale, p_drop, layer)
        ctx.holder = holder
        if need_bwd:
            ctx.save_for_backward(packs[0], packs[1], O, lse, mask_u8, keep)
        return O

    @staticmethod
    @K.backward_scope
    def backward(ctx, dO):
        Qf, Q16, O, lse, mask_u8, keep = ctx.saved_tensors
        B, Lq, S, H, dk, dh, nch, scale, p_drop, layer = ctx.meta
        h = ctx.holder
        L = h.dims[0]
        d = H * dh
        dO = dO.contiguous()
        dO4 = dO.view(B, Lq, H, dh)
        D = K.rowdot(dO4, O.contiguous().view(B, Lq, H, dh))                             # [B,H,Lq] = rowsum(dO . O)
        dOf, dO16 = K.attn_pack_multi([(dO4, 1.0, 322), (dO4, 1.0, 16)])
        dq, dk_, dv_ = K.mha_bwd(Qf, h.Kf[layer], h.Vf[layer], dOf, h.K16[layer], Q16, dO16, mask_u8, lse, D, keep, B, H, Lq, S, dk, dh, nch,
                                 scale, p_drop)
        if not ctx.needs_input_grad[1]:                 # the memory side takes no gradient (frozen projections, detached memory): nothing to scatter
            return dq, None, None, None, None, None, None
        if h.dYm is None:
            h.dYm = torch.empty((B * S, 2 * L * d), device=dO.device, dtype=torch.bfloat16)
            h.dYp = torch.empty((B * S, L * d), device=dO.device, dtype=torch.bfloat16)
            h.written = set()
        if layer in h.written:
            # spe_kv_grad_scatter stores (it does not accumulate): a second cross-attention node on the same (holder, layer) in one graph - a
            # mem_cache reused across decoder passes - would silently drop the first node's key / value gradients
            raise RuntimeError(f"MemoryKV: the key / value gradients of decoder layer {layer} were already written in this backward pass; a "
                               "(holder, layer) pair feeds ONE cross-attention node per graph (do not reuse a mem_cache that holds MemoryKV entries)")
        K.kv_grad_scatter(dk_, dv_, h.dYm, h.dYp, layer, B, S, H, dh)
        h.written.add(layer)
        return dq, h.ztok, None, None, None, None, None


def memory_side_kv_ok(memory, Wm, Wp, H):
    """The fragment path of the decoder's memory side: benchmark precision modes, flash-MHA shapes, GEMM alignment."""
    B, S, d = memory.shape
    dh = d // H
    return (FLASH_MHA and MEMKV and memory.is_cuda and K.get_precision() != "bf16x3" and B * S >= 2048 and S >= FLASH_MIN_KEYS and d % 64 == 0
            and dh % 8 == 0 and dh <= 48 and len(Wm) <= 32 and all(W.shape == (d, d) and W.is_contiguous() for W in list(Wm) + list(Wp)))


def memory_side_kv(memory, pos, H, Wm, Wp, bm, bp):
    """-> (holder, tokens): see _MemorySideKV.  Wm = [kcontent_0, v_0, kcontent_1, v_1, ...], Wp = [kpos_0, ...]."""
    holder = MemoryKV()
    holder.want_bwd = torch.is_grad_enabled()
    toks = _MemorySideKV.apply(holder, memory, pos, H, *Wm, *Wp, *bm, *bp)
    return holder, toks


def cross_attention_kv(q, tok, holder, layer, key_padding_mask, scale, p_drop):
    # the uint8 image of the padding mask is the same for every layer of a decoder pass: converted once, kept on the holder
    m = None
    if key_padding_mask is not None:
        ent = getattr(holder, "mask_u8", None)
        if ent is None or ent[0] is not key_padding_mask:
            ent = holder.mask_u8 = (key_padding_mask, key_padding_mask.to(torch.uint8).contiguous())
        m = ent[1]
    return _CrossAttentionKV.apply(q, tok, holder, layer, m, float(scale), float(p_drop))


MEMKV = True
FLASH_MHA = True       # module attribute (tests and tools/error_budget.py flip it), not an environment knob
FLASH_MIN_KEYS = 512


def attention(q, k, v, key_padding_mask=None, scale=1.0, p_drop=0.0, need_map=False):
    """key_padding_mask: bool/uint8 [B,Lk], True = padded key."""
    m = None
    if key_padding_mask is not None:
        m = key_padding_mask.to(torch.uint8).contiguous()
    # flash kernels when no map is wanted and the key axis is long (decoder self-attention over 100 queries is cheaper
    # through the three small materialising launches)
    if (FLASH_MHA and not need_map and K.get_precision() != "bf16x3" and q.is_cuda and q.shape[3] <= 96 and v.shape[3] <= 64
            and k.shape[1] >= FLASH_MIN_KEYS and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1):
        return _AttentionFlash.apply(q, k, v, m, float(scale), float(p_drop)), None
    return _Attention.apply(q, k, v, m, float(scale), float(p_drop), need_map)


# ---------------------------------------------------------------------------------------------
class _PatchEmbed(Function):
    """Conv2d(3,C,16,16) patch embedding as gather + GEMM (reference cait.py:518-528)."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, img, W, b, P):
        B, Cin, Hi, Wi = img.shape
        cols = K.patchify(img.contiguous(), P)
        W2 = W.reshape(W.shape[0], -1)
        y, _, xsave = K.linear_fwd(cols, W2, b, save_for_dw=ctx.needs_input_grad[1])
        ctx.save_for_backward(xsave, W2)
        ctx.wshape = W.shape
        return y.view(B, (Hi // P) * (Wi // P), W.shape[0])

    @staticmethod
    @K.backward_scope
    def backward(ctx, dy):
        xsave, W2 = ctx.saved_tensors
        dy2 = dy.reshape(-1, W2.shape[0]).contiguous()
        _, dW, db = K.linear_bwd(dy2, xsave, W2, need_dx=False)
        return None, dW.view(ctx.wshape), db, None


def patch_embed(img, W, b, P):
    return _PatchEmbed.apply(img, W, b, P)


class _AddRows(Function):
    """x [B,N,C] + table [N,C] (pos-embed add, cait.py:623-624)."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, x, table):
        ctx.B = x.shape[0]
        return K.add_rows(x.contiguous(), table.contiguous())

    @staticmethod
    @K.backward_scope
    def backward(ctx, dy):
        dy = dy.contiguous()
        dt = None
        if ctx.needs_input_grad[1]:
            n = dy.numel() // ctx.B
            dt = K.colsum(dy.view(ctx.B, n)).view(dy.shape[1:])
        return dy, dt


def add_rows(x, table):
    return _AddRows.apply(x, table)


class _Add(Function):
    """a + b for equally shaped activations (residual / positional adds), one float4 pass."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        return K.add_rows(a, b)

    @staticmethod
    @K.backward_scope
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    assert a.shape == b.shape
    return _Add.apply(a, b)


class _BicubicGrid(Function):
    """pos_embed [1, gh*gw, C] -> [1, h*w, C] (cait.py:598-613)."""

    @staticmethod
    @K.forward_scope
    def forward(ctx, pe, gh, gw, h, w):
        ctx.g = (gh, gw, h, w)
        return K.bicubic(pe[0].contiguous(), gh, gw, h, w).unsqueeze(0)

    @staticmethod
    @K.backward_scope
    def backward(ctx, dy):
        gh, gw, h, w = ctx.g
        return K.bicubic(dy[0].contiguous(), gh, gw, h, w, backward=True).unsqueeze(0), None, None, None, None


def bicubic_grid(pe, gh, gw, h, w):
    return _BicubicGrid.apply(pe, gh, gw, h, w)
