"""Round-6 GPU tests.

The two backward kernels of the talking-heads attention (csrc/attn_flash_bwd.hip: spe_talking_bwdk_pass1 = key-major pass 1 + dV,
spe_talking_bwdq_pass2 = query-major pass 2 + dQ) DIRECTLY against an fp64 restatement of the autograd of reference models/cait.py:377-389,
through the C-ABI, at the token counts of cfg2 (2 x 4150) and cfg5 (1 x 6200) and on ragged / small shapes, with and without attention
dropout: D, dS, dQ, dV, dWl, dbl, dWw, dbw.  (Rounds 4 / 5 tested them against the kernels they replaced; those are gone.)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def _dense_from_blocks(T, N):
    """[B,H,nt,nt,64,4] blocked scores -> dense [B,H,N,N]: lane l of block (qt,kt) = query qt*16+(l&15), keys kt*16+4*(l>>4)+i."""
    B, H, nt = T.shape[:3]
    return T.view(B, H, nt, nt, 4, 16, 4).permute(0, 1, 2, 5, 3, 4, 6).reshape(B, H, nt * 16, nt * 16)[:, :, :N, :N]


def _keep_from_bits(bits, H, N):
    """The flash forward's keep flags [B,nt,nt,64] int32 -> dense bool [B,H,N,N]: lane l of tile (qt,kt) holds query qt*16+(l&15); bit
    hp*8 + 2r + e is key kt*16 + 4*(l>>4) + r of head 2hp + e (csrc/attn_flash.hip, csrc/attn_flash_bwd.hip)."""
    B, nt = bits.shape[:2]
    x = bits.view(B, nt, nt, 4, 16)                                   # [b, qt, kt, lane group, lane in group]
    out = torch.empty((B, H, nt, 16, nt, 4, 4), device=bits.device, dtype=torch.bool)        # [b, h, qt, lq, kt, lg, r]
    for hp in range(H // 2):
        for e in range(2):
            for r in range(4):
                bit = ((x >> (hp * 8 + 2 * r + e)) & 1).bool()          # [b, qt, kt, lg, lq]
                out[:, 2 * hp + e, :, :, :, :, r] = bit.permute(0, 1, 4, 2, 3)
    return out.reshape(B, H, nt * 16, nt * 16)[:, :, :N, :N]


def _run_kernels(B, H, N, dh, p_drop, dev, seed=1):
    """Pack the operands the way the attention node does, run the statistics pass and the flash forward (for the keep flags), then the two backward
    kernels.  -> dict of inputs (fp32 originals) and kernel results."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(seed)
    C = H * dh
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dev)
    Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bl = (0.1 * torch.randn(H, generator=g)).to(dev)
    Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bw = (0.1 * torch.randn(H, generator=g) / N).to(dev)
    dO = torch.randn(B, N, C, generator=g).to(dev)
    scale = dh ** -0.5
    v5 = qkv.view(B, N, 3, H, dh)
    q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
    Qf, Kf, V16, Vf, K16 = K.attn_pack_multi([(q, scale * K.LOG2E, 32 + K.F16), (k, 1.0, 32 + K.F16), (v, 1.0, 16 + K.F16), (v, 1.0, 32), (k, 1.0, 16)])
    dO4 = dO.view(B, N, H, dh)
    dOf, dO16 = K.attn_pack_multi([(dO4, 1.0, 32), (dO4, 1.0, 16)])
    nt = (N + 15) // 16
    spw0, _ = K.fused_plan(B, N, 0)
    ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
    K.talking_stats(Qf, Kf, Wl, bl, ws, B, H, N, dh)
    M, IL, c0 = K.attn_merge_rows(ws, bl, B, H, N, spw0)
    bits = K.talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, c0, B, H, N, dh, p_drop, 7, 3, want_bits=True)[3] if p_drop > 0 else None

    def run():
        dv = torch.full((B, N, H, dh), float("nan"), device=dev)
        dv16 = torch.zeros(B, N, H, dh, device=dev, dtype=torch.bfloat16)
        Drows, ws_w = K.talking_bwdk_pass1(Qf, dOf, dO16, Kf, Vf, Wl, Ww, bw, c0, bits, dv, dv16, B, H, N, dh, p_drop)
        dS = K.score_blocks(B, H, N, dev)
        dq = torch.full((B, N, H, dh), float("nan"), device=dev)
        dq16 = torch.zeros(B, N, H, dh, device=dev, dtype=torch.bfloat16)
        K.talking_bwdq_pass2(Qf, dOf, Kf, Vf, K16, Wl, Ww, c0, Drows, ws_w, dS, dq, dq16, scale, bits, B, H, N, dh, p_drop)
        w = K.talking_wgrad_reduce(ws_w, H, (None, None, None, None))
        return dict(Drows=Drows, dv=dv, dv16=dv16, dS=dS, dq=dq, dq16=dq16, dWl=w[0], dbl=w[1], dWw=w[2], dbw=w[3])

    return dict(q=q, k=k, v=v, dO=dO4, Wl=Wl, bl=bl, Ww=Ww, bw=bw, scale=scale, bits=bits, run=run)


def _fp64_backward(x, b, H, N, p_drop):
    """fp64 autograd of cait.py:377-389 for image b, written out (S -> S' = Wl S + bl -> P = softmax -> P' = Ww P + bw -> dropout -> O = P'd V), on the
    operands AS THE KERNELS SEE THEM (q scale log2 e and k rounded to fp16 for the scores; v, dO and the k of the dQ product rounded to bf16):
    what remains is the kernels' internal arithmetic (fp16 / bf16 packs of P, dP', dS', dS in front of the matrix instructions)."""
    from spe_amd import kernels as K
    f = lambda t: t[b].permute(1, 0, 2)                                  # [N,H,dh] -> [H,N,dh]
    qh = (f(x["q"]) * (x["scale"] * K.LOG2E)).half().double() / K.LOG2E    # = scale q
    kh = f(x["k"]).half().double()
    kb = f(x["k"]).bfloat16().double()
    vb = f(x["v"]).bfloat16().double()
    dOb = f(x["dO"]).bfloat16().double()
    Wl, bl, Ww, bw = (x[n].double() for n in ("Wl", "bl", "Ww", "bw"))
    S = qh @ kh.transpose(1, 2)                                          # [H,N,N]
    P = (torch.einsum("gh,hqk->gqk", Wl, S) + bl[:, None, None]).softmax(-1)
    keep = None
    if p_drop > 0:
        keep = _keep_from_bits(x["bits"][b:b + 1], H, N)[0].double() / (1.0 - p_drop)
    dPp = dOb @ vb.transpose(1, 2)                                       # dP'd
    Pp = torch.einsum("gh,hqk->gqk", Ww, P) + bw[:, None, None]
    if keep is not None:
        Pp *= keep
        dPp *= keep
    dV = Pp.transpose(1, 2) @ dOb                                        # [H,N,dh]
    del Pp
    dWw = torch.einsum("gqk,hqk->gh", dPp, P)
    dbw = dPp.sum((1, 2))
    dP = torch.einsum("gh,gqk->hqk", Ww, dPp)
    del dPp
    D = (dP * P).sum(-1)                                                 # [H,N]
    dSp = P * (dP - D[:, :, None])
    del dP, P
    dWl = torch.einsum("gqk,hqk->gh", dSp, S)
    dbl = dSp.sum((1, 2))
    dS = torch.einsum("gh,gqk->hqk", Wl, dSp)
    del dSp, S
    dQ = x["scale"] * (dS @ kb)                                          # [H,N,dh]
    return dict(D=D, dS=dS, dQ=dQ, dV=dV, dWl=dWl, dbl=dbl, dWw=dWw, dbw=dbw, keep_rate=None if keep is None else float((keep > 0).double().mean()))


CASES = [(2, 8, 4150, 48, 0.0), (2, 8, 4150, 48, 0.05), (1, 8, 6200, 48, 0.0), (1, 8, 6200, 48, 0.1),
         (1, 8, 100, 48, 0.0), (2, 4, 196, 48, 0.0), (2, 8, 1100, 48, 0.1), (1, 4, 300, 32, 0.05), (2, 8, 400, 16, 0.0), (1, 8, 2070, 48, 0.0),
         (1, 4, 64, 64, 0.2)]


@pytest.mark.parametrize("B,H,N,dh,p_drop", CASES)
def test_attention_backward_kernels_vs_fp64(dev, B, H, N, dh, p_drop):
    """spe_talking_bwdk_pass1 and spe_talking_bwdq_pass2 against the fp64 restatement: D (query-major rows, zero beyond N), dV and its bf16
    copy, the dS blocks, dQ and its bf16 copy, and - through spe_talking_wgrad_reduce - dWl, dbl (exact value 0: softmax is shift invariant),
    dWw, dbw; with dropout the keep flags are the ones the flash forward stored (rate checked).  Bitwise reproducible run to run."""
    from spe_amd import kernels as K
    if not K.fused_supported(H, dh):
        pytest.skip("shape not on the fused attention path")
    x = _run_kernels(B, H, N, dh, p_drop, dev)
    r = x["run"]()
    torch.cuda.synchronize()
    assert all(torch.isfinite(r[n].float()).all() for n in ("Drows", "dv", "dq", "dWl", "dbl", "dWw", "dbw"))
    assert (r["Drows"][:, N:] == 0).all()                                # pass 2 reads the padded rows of the last q-tile
    assert torch.equal(r["dv16"].float(), r["dv"].to(torch.bfloat16).float()) and torch.equal(r["dq16"].float(), r["dq"].to(torch.bfloat16).float())
    dSd = _dense_from_blocks(r["dS"], N)
    acc = {n: 0.0 for n in ("dWl", "dbl", "dWw", "dbw")}
    errs = {"D": 0.0, "dS": 0.0, "dQ": 0.0, "dV": 0.0}
    for b in range(B):                                                   # per image: bounds the fp64 N x N temporaries (~1.1 GB each at N = 4150)
        ref = _fp64_backward(x, b, H, N, p_drop)
        errs["D"] = max(errs["D"], rel(r["Drows"][b, :N].t(), ref["D"]))
        errs["dS"] = max(errs["dS"], rel(dSd[b], ref["dS"]))
        errs["dQ"] = max(errs["dQ"], rel(r["dq"][b].permute(1, 0, 2), ref["dQ"]))
        errs["dV"] = max(errs["dV"], rel(r["dv"][b].permute(1, 0, 2), ref["dV"]))
        for n in acc:
            acc[n] = acc[n] + ref[n]
        if p_drop > 0:
            nel = H * N * N
            assert abs((1.0 - ref["keep_rate"]) - p_drop) < 0.01 + 4.0 * (p_drop * (1 - p_drop) / nel) ** 0.5, ref["keep_rate"]
        del ref
    for n in ("dWl", "dWw", "dbw"):
        errs[n] = rel(r[n], acc[n])
    errs["dbl"] = float(r["dbl"].abs().max() / r["dWl"].abs().max())
    print(f"[attention backward vs fp64 B={B} H={H} N={N} dh={dh} p={p_drop}] " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    # measured (round 6): see profiles/r06_attn_bwd_fp64.txt; the bounds are ~2x the worst case.  dS is stored as bf16 (2^-9 per element)
    assert errs["D"] < 3e-3 and errs["dV"] < 4e-3 and errs["dQ"] < 6e-3 and errs["dS"] < 6e-3, errs
    tolw = 6e-3 if N >= 256 else 1.2e-2                    # a 64 x 64 problem has too few bf16-rounded terms per weight gradient to average the rounding out
    assert errs["dWl"] < tolw and errs["dWw"] < tolw and errs["dbw"] < tolw, errs
    assert errs["dbl"] < 2e-2, errs
    r2 = x["run"]()
    torch.cuda.synchronize()
    for n in ("Drows", "dv", "dq", "dWl", "dbl", "dWw", "dbw"):
        assert torch.equal(r[n], r2[n]), n
    assert torch.equal(r["dS"].view(torch.int16), r2["dS"].view(torch.int16))


def test_attention_node_has_one_backward_composition(dev):
    """The product has ONE backward composition of the talking-heads attention (no SPE_BWDQ switch, no round-3 passes): the node's gradients against
    the fp64 restatement with attention dropout on (flags from the flash forward), and the library exports none of the retired entries."""
    from spe_amd import kernels as K, lib, ops
    assert not hasattr(ops, "BWDQ_MODE") and not hasattr(ops, "BWDQ")
    for gone in ("spe_talking_flash_dv", "spe_talking_bwdq_pass1", "spe_talking_fused", "spe_talking_fused_bits"):
        assert gone not in lib.PROTOS, gone
    K.set_precision("bf16s")
    g = torch.Generator().manual_seed(11)
    B, H, N, dh, p = 2, 8, 700, 48, 0.1
    C = H * dh
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dev).requires_grad_(True)
    Wl = (torch.eye(H) + 0.2 * torch.randn(H, H, generator=g)).to(dev).requires_grad_(True)
    Ww = (torch.eye(H) + 0.2 * torch.randn(H, H, generator=g)).to(dev).requires_grad_(True)
    bl = (0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_(True)
    bw = (0.1 + 0.02 * torch.randn(H, generator=g)).to(dev).requires_grad_(True)
    w = torch.randn(B, N, C, generator=g).to(dev)
    K.manual_seed(31)
    O = ops.talking_heads_attention(qkv, Wl, bl, Ww, bw, H, dh ** -0.5, p)
    grads = torch.autograd.grad((O * w).sum(), (qkv, Wl, Ww, bw))
    # the keep mask: O is linear in V, so re-run the forward with the same stream on the node's own fragments is not needed - recover the mask
    # from the flags of an identical flash forward (same seed / offset the node drew)
    with torch.no_grad():
        v5 = qkv.detach().view(B, N, 3, H, dh)
        Qf, Kf, V16 = K.attn_pack_multi([(v5[:, :, 0], dh ** -0.5 * K.LOG2E, 32 + K.F16), (v5[:, :, 1], 1.0, 32 + K.F16), (v5[:, :, 2], 1.0, 16 + K.F16)])
        nt = (N + 15) // 16
        ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
        K.talking_stats(Qf, Kf, Wl.detach(), bl.detach(), ws, B, H, N, dh)
        _, _, c0 = K.attn_merge_rows(ws, bl.detach(), B, H, N, K.fused_plan(B, N, 0)[0])
        bits = K.talking_flash_fwd(Qf, Kf, V16, Wl.detach(), Ww.detach(), bw.detach(), c0, B, H, N, dh, p, 31, 1, want_bits=True)[3]
        keep = _keep_from_bits(bits, H, N).double() / (1.0 - p)
    dd = [t.detach().double().requires_grad_() for t in (qkv, Wl, bl, Ww, bw)]
    q, k, v = dd[0].reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    attn = (q * dh ** -0.5) @ k.transpose(-2, -1)
    attn = torch.nn.functional.linear(attn.permute(0, 2, 3, 1), dd[1], dd[2]).permute(0, 3, 1, 2).softmax(-1)
    attn = torch.nn.functional.linear(attn.permute(0, 2, 3, 1), dd[3], dd[4]).permute(0, 3, 1, 2) * keep
    ref = (attn @ v).transpose(1, 2).reshape(B, N, C)
    rg = torch.autograd.grad((ref * w.double()).sum(), (dd[0], dd[1], dd[3], dd[4]))
    assert rel(O, ref) < 2e-3, rel(O, ref)
    for nm, a, b_ in zip(("dqkv", "dWl", "dWw", "dbw"), grads, rg):
        assert rel(a, b_) < 1e-2, (nm, rel(a, b_))


# ------------------------------------------------------------------------------------------------ data parallel: deferred sums at world 2
def _dp2_defer_worker(rank, world, port, out):
    """One of two processes on the SAME GPU (gloo moves the buckets through the host; RCCL refuses two ranks on one device): the product model
    in the benchmark precision with a backbone wide enough for the bf16-copy GEMM path (the producers of deferred sums), per-rank data, the
    bucketed all-reduce beside the backward - with the deferred bias / LayerNorm / LayerScale sums ON and OFF."""
    import os
    import sys
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch.nn as nn
        from spe_amd import kernels as K
        from spe_amd.dp import GradAllReducer
        from spe_amd.models.cait import LayerScale_Block
        from spe_amd.models.layers import LayerNorm
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        K.set_precision("bf16s")
        torch.manual_seed(7)                                  # identical replicas
        C, H, B, N = 384, 8, 1, 2100

        class Net(nn.Module):
            def __init__(self):
                super().__init__()
                self.blocks = nn.ModuleList([LayerScale_Block(C, H, init_values=0.3) for _ in range(3)])
                self.norm = LayerNorm(C)

            def forward(self, x):
                for b in self.blocks:
                    x = b(x)
                return self.norm(x)

        net = Net().to(dev).train()
        g = torch.Generator().manual_seed(100 + rank)        # per-rank data
        x = torch.randn(B, N, C, generator=g).to(dev)
        w = torch.randn(B, N, C, generator=g).to(dev)
        params = [p for p in net.parameters() if p.requires_grad]
        names = [n for n, p in net.named_parameters() if p.requires_grad]

        def run(defer):
            old = K.DEFER_REDUCE
            K.DEFER_REDUCE = defer
            try:
                # small buckets: several all-reduces go out DURING the backward, each after the flush of the sums that belong in it
                red = GradAllReducer(params, bucket_bytes=1 << 20, flatten_params=False)
                assert red.collective and red.world == 2
                res, pend = [], []
                for _ in range(3):
                    red.reset()
                    (net(x) * w).sum().backward()
                    pend.append(K.lib.load().spe_reduce_pending())
                    red.finish()
                    torch.cuda.synchronize()
                    res.append(torch.cat([p.grad.flatten() for p in params]).clone())
                deferring = red._defer
                red.remove()
                return res, deferring, pend
            finally:
                K.DEFER_REDUCE = old

        tree, d0, _ = run(False)
        deff, d1, pend = run(True)
        assert not d0 and d1, (d0, d1)
        assert all(p == 0 for p in pend), pend                # nothing is left pending when backward() returns
        for step in (1, 2):                                   # deferral starts after the first step
            a, b = deff[step], tree[step]
            # the deferred flush and the per-launch tree add the same fp32 partials in different orders: rounding-level differences only
            offs = 0
            for n_, p_ in zip(names, params):
                k = p_.numel()
                da, db = a[offs:offs + k], b[offs:offs + k]
                assert (da - db).norm() <= 4e-6 * db.norm() + 1e-7, (n_, step, float((da - db).norm() / (db.norm() + 1e-30)))
                offs += k
        assert torch.equal(deff[1], deff[2])                  # same inputs, same weights: bitwise reproducible through the collective
        # ... and the two ranks hold bitwise the same reduced gradients (every bucket went out after its deferred sums were flushed on both)
        both = [torch.zeros_like(deff[2]) for _ in range(world)]
        dist.all_gather(both, deff[2])
        assert torch.equal(both[0], both[1])
        # the reduced gradient is the SUM of the ranks' local gradients (the reducer leaves the sum; FlatAdamW folds 1 / world into its launch)
        loc = torch.autograd.grad((net(x) * w).sum(), params)
        flat = torch.cat([t.flatten() for t in loc])
        dist.all_reduce(flat)
        scale = 1.0 if float((deff[2] - flat).norm()) < float((deff[2] * world - flat).norm()) else float(world)
        assert (deff[2] * scale - flat).norm() <= 2e-5 * flat.norm(), float((deff[2] * scale - flat).norm() / flat.norm())
        out[rank] = True
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_dp_world2_deferred_sums_two_processes_one_gpu(dev):
    """ADVICE r4 / VERDICT r5 item 8: the deferred bias / LayerNorm / LayerScale sums (csrc/det_reduce.h, spe_reduce_defer_*: partial rows left in an arena,
    ONE flush launch per group of producers, flushed before the bucket they belong to is all-reduced) at world 2 with the bucketed all-reduce running
    beside the backward: equal to the per-launch tree sums to fp32 summation order (4e-6), nothing pending after backward(), bitwise reproducible step to
    step, bitwise identical on both ranks, and equal to the all-reduced local gradients.  Reference: DDP's gradient all-reduce, main.py:171-173."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp2_defer_worker, args=(2, port, out), nprocs=2, join=True)
    assert out.get(0) and out.get(1)


# ------------------------------------------------------------------------------------------------ glue diet
@pytest.mark.parametrize("R,C,p", [(400, 384, 0.1), (8300, 384, 0.0), (77, 64, 0.3)])
def test_two_output_residual_layer_norm(dev, R, C, p):
    """ops.res_drop_layer_norm2 - (y, y) of norm(x + dropout(z)) for a post-norm layer whose output feeds a Linear AND the next residual site
    (reference models/transformer.py:384-427, 279-287); the two gradients are added inside the LayerNorm backward kernel (spe_layernorm_res_bwd dy2) -
    against the one-output node with the SUM of the gradients (same Philox stream: same mask): bitwise the same y, gradients to fp32 rounding; one
    consumer only (the other gradient None) in both positions; p = 0 against fp64."""
    from spe_amd import kernels as K, ops
    g_ = torch.Generator().manual_seed(R + C + 1)
    x = torch.randn(1, R, C, generator=g_).to(dev)
    z = torch.randn(x.shape, generator=g_).to(dev)
    w = (1 + 0.2 * torch.randn(C, generator=g_)).to(dev); b = (0.1 * torch.randn(C, generator=g_)).to(dev)
    ga = torch.randn(x.shape, generator=g_).to(dev); gb = torch.randn(x.shape, generator=g_).to(dev)

    def run(two, g1, g2):
        xs, zs, ws, bs = (t.clone().requires_grad_() for t in (x, z, w, b))
        K.manual_seed(99)
        if two:
            y1, y2 = ops.res_drop_layer_norm2(xs, zs, ws, bs, 1e-5, p, True)
            assert y1.data_ptr() == y2.data_ptr()
            outs = [(y1, g1), (y2, g2)]
            loss = sum((y * g).sum() for y, g in outs if g is not None)
            y = y1
        else:
            y = ops.res_drop_layer_norm(xs, zs, ws, bs, 1e-5, p, True)
            loss = (y * ((g1 if g1 is not None else 0) + (g2 if g2 is not None else 0))).sum()
        return (y.detach(),) + torch.autograd.grad(loss, (xs, zs, ws, bs))

    for g1, g2 in ((ga, gb), (ga, None), (None, gb)):
        r2, r1 = run(True, g1, g2), run(False, g1, g2)
        assert torch.equal(r2[0], r1[0])
        for a, c in zip(r2[1:], r1[1:]):
            assert rel(a, c) < 2e-6, rel(a, c)
    if p == 0.0:
        xd, zd, wd_, bd = (t.double().requires_grad_() for t in (x, z, w, b))
        yd = torch.nn.functional.layer_norm(xd + zd, (C,), wd_, bd, 1e-5)
        gd = torch.autograd.grad(yd, (xd, zd, wd_, bd), (ga + gb).double())
        for a, c in zip(run(True, ga, gb), (yd,) + gd):
            assert rel(a, c) < 1e-5


def test_rowdot_kernel(dev):
    """spe_rowdot: D[b][h][q] = sum_d dO[b][q][h][d] O[b][q][h][d] (the softmax backward's row term of the decoder's cross attention) against torch."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    for B, L, H, dh in ((2, 200, 8, 48), (1, 7, 4, 10), (3, 4150, 8, 32)):
        x = torch.randn(B, L, H, dh, generator=g).to(dev); y = torch.randn(B, L, H, dh, generator=g).to(dev)
        D = K.rowdot(x, y)
        ref = (x.double() * y.double()).sum(-1).permute(0, 2, 1)
        assert D.shape == (B, H, L) and rel(D, ref) < 1e-6


def test_group_linear_with_addends(dev):
    """ops.group_linear(x, mods, adds=...): output i + adds[i] from the same launch (the decoder's q = sa_qcontent_proj(tgt) + sa_qpos_proj(query_pos),
    k likewise, reference models/transformer.py:368-374) against the group launch followed by torch adds: outputs to fp32 rounding (the addend joins the
    accumulator before the bias instead of after it), every gradient - input, weights, biases, addends - bitwise."""
    from spe_amd import kernels as K, ops
    from spe_amd.models.layers import Linear
    K.set_precision("bf16s")
    torch.manual_seed(5)
    R, d = 400, 384
    mods = [Linear(d, d).to(dev) for _ in range(3)]
    x0 = torch.randn(2, R // 2, d, device=dev)
    a0 = [torch.randn(2, R // 2, d, device=dev), torch.randn(2, R // 2, d, device=dev), None]
    ws = [torch.randn(2, R // 2, d, device=dev) for _ in range(3)]

    def run(fused):
        x = x0.clone().requires_grad_(True)
        adds = [None if a is None else a.clone().requires_grad_(True) for a in a0]
        for m in mods:
            m.weight.grad = m.bias.grad = None
        if fused:
            ys = ops.group_linear(x, mods, adds=adds)
        else:
            ys = [y if a is None else y + a for y, a in zip(ops.group_linear(x, mods), adds)]
        sum((y * w).sum() for y, w in zip(ys, ws)).backward()
        return ([y.detach() for y in ys], x.grad, [m.weight.grad.clone() for m in mods], [m.bias.grad.clone() for m in mods],
                [None if a is None else a.grad for a in adds])

    f, c = run(True), run(False)
    for yf, yc in zip(f[0], c[0]):
        assert rel(yf, yc) < 1e-6
    assert torch.equal(f[1], c[1])
    for i in range(3):
        assert torch.equal(f[2][i], c[2][i]) and torch.equal(f[3][i], c[3][i])
    assert torch.equal(f[4][0], c[4][0]) and torch.equal(f[4][1], c[4][1]) and f[4][2] is None
