"""Round-4 GPU tests of the host-side contracts the advisor flagged: accumulate-into gradients larger than the
reducer's pre-zeroed views, forward precision of a block recomputed inside a backward (torch.utils.checkpoint), launches from a
second stream next to the shared reduction workspace."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_wide_bias_gradient_does_not_accumulate_across_steps(dev):
    """A bias of > 16384 elements (a wide class / vocabulary head) is an ACCUMULATE-INTO gradient (column sums added onto the
    bucket view) that GradAllReducer.reset() does not pre-zero from the second step on (dp.py: ZERO_MAX): kernels._zeros_or zeroes
    such a view at hand-out time.  Three steps with different inputs, a parameter that fires only in step 1 included: every
    step's gradients equal the fp64 reference of THAT step (not the running sum).  Reference: the autograd of nn.Linear at
    models/conditional_detr.py:104-110 under DDP (main.py:172), whose buckets start every backward from zero."""
    from spe_amd import kernels as K, ops
    from spe_amd.dp import GradAllReducer
    K.set_precision("bf16x3")
    g = torch.Generator().manual_seed(5)
    Nout, Kin, R = 20000, 64, 256
    W = torch.nn.Parameter((0.05 * torch.randn(Nout, Kin, generator=g)).to(dev))
    b = torch.nn.Parameter(torch.zeros(Nout, device=dev))
    W2 = torch.nn.Parameter((0.05 * torch.randn(Nout, Kin, generator=g)).to(dev))          # used in step 0 only
    b2 = torch.nn.Parameter(torch.zeros(Nout, device=dev))
    red = GradAllReducer([W, b, W2, b2])
    assert b.numel() > red.ZERO_MAX == K.ACC_ZERO_MAX
    try:
        for step in range(3):
            x = torch.randn(R, Kin, generator=g).to(dev)
            go = torch.randn(R, Nout, generator=g).to(dev)
            red.reset()
            y = ops.linear(x, W, b)
            if step == 0:
                y = y + ops.linear(x, W2, b2)
            (y * go).sum().backward()
            red.finish()
            db_ref = go.double().sum(0)
            dW_ref = go.double().t() @ x.double()
            assert (b.grad.double() - db_ref).abs().max() <= 1e-4 * db_ref.abs().max(), step
            assert (W.grad.double() - dW_ref).abs().max() <= 1e-4 * dW_ref.abs().max(), step
            if step == 0:
                assert (b2.grad.double() - db_ref).abs().max() <= 1e-4 * db_ref.abs().max()
            else:       # no gradient this step: zeros, like DDP's unused-parameter path - not step 0's leftovers
                assert float(b2.grad.abs().max()) == 0.0 and float(W2.grad.abs().max()) == 0.0, step
    finally:
        red.remove()


def test_checkpointed_block_recomputes_at_forward_precision(dev):
    """bf16s runs forward products on split operands and backward products on single bf16 operands; which one a launch gets is a
    thread-local flag set around every backward (kernels.backward_scope).  torch.utils.checkpoint(use_reentrant=False) recomputes
    the block INSIDE the wrapped backward: kernels.forward_scope makes that recomputation a forward again, so the recomputed
    activations - and with them every gradient - are bitwise those of the run without checkpointing."""
    from torch.utils.checkpoint import checkpoint
    from spe_amd import kernels as K, ops
    K.set_precision("bf16s")
    g = torch.Generator().manual_seed(9)
    C, R = 384, 2304
    W1 = (0.05 * torch.randn(4 * C, C, generator=g)).to(dev).requires_grad_(); b1 = torch.zeros(4 * C, device=dev, requires_grad=True)
    W2 = (0.05 * torch.randn(C, 4 * C, generator=g)).to(dev).requires_grad_(); b2 = torch.zeros(C, device=dev, requires_grad=True)
    gam = torch.ones(C, device=dev, requires_grad=True); bet = torch.zeros(C, device=dev, requires_grad=True)
    x = torch.randn(1, R, C, generator=g).to(dev).requires_grad_()
    go = torch.randn(1, R, C, generator=g).to(dev)

    def block(t):
        return t + ops.mlp_gelu(ops.layer_norm(t, gam, bet, 1e-6), W1, b1, W2, b2)

    outs = []
    for use_ckpt in (False, True):
        y = checkpoint(block, x, use_reentrant=False) if use_ckpt else block(x)
        grads = torch.autograd.grad(y, (x, W1, b1, W2, gam), go)
        outs.append([y.detach().clone()] + [t.clone() for t in grads])
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)


def test_launch_from_second_stream_is_ordered_behind_the_first(dev):
    """The deterministic reductions share one ticket / slab workspace per process (csrc/det_reduce.h): kernels._call orders a launch
    from another stream behind everything the previous stream was given.  Column sums issued alternately from two streams stay
    bitwise equal to the single-stream result; a launch on another DEVICE index is refused."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    x = torch.randn(8300, 1536, generator=g).to(dev)
    ref = K.colsum(x, torch.empty(1536, device=dev), accumulate=False).clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    outs = []
    for t in range(8):
        if t % 2:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                outs.append(K.colsum(x, torch.empty(1536, device=dev), accumulate=False))
        else:
            outs.append(K.colsum(x, torch.empty(1536, device=dev), accumulate=False))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)


@pytest.mark.gpu
def test_group_linear_matches_separate_linears():
    """ops.group_linear (one launch each way for several Linears on one input: the decoder's query-side projections, reference
    models/transformer.py:368-372, 399) against the same Linears evaluated one by one: outputs, weight and bias gradients bitwise
    (same per-tile program), input gradient to rounding (one contraction over all blocks instead of a sum of n); an output
    that receives no gradient leaves its parameters' gradients untouched."""
    import torch.nn as nn
    from spe_amd import kernels as K, ops
    from spe_amd.models.layers import Linear
    dev = torch.device("cuda:0")
    K.set_precision("bf16s")
    torch.manual_seed(3)
    for n, R, d in ((3, 400, 384), (13, 400, 384), (2, 300, 256)):
        mods = [Linear(d, d).to(dev) for _ in range(n)]
        x = torch.randn(2, R // 2, d, device=dev, requires_grad=True)
        assert ops.group_linear_ok(x, mods)
        ws = [torch.randn(2, R // 2, d, device=dev) for _ in range(n)]
        skip = n - 1 if n > 2 else None                     # this output is left out of the loss
        ys = ops.group_linear(x, mods)
        loss = sum((y * w).sum() for i, (y, w) in enumerate(zip(ys, ws)) if i != skip)
        loss.backward()
        got = (x.grad.clone(), [m.weight.grad for m in mods], [m.bias.grad for m in mods], [y.detach() for y in ys])
        x.grad = None
        for m in mods:
            m.weight.grad = m.bias.grad = None
        ys2 = [m(x) for m in mods]
        loss2 = sum((y * w).sum() for i, (y, w) in enumerate(zip(ys2, ws)) if i != skip)
        loss2.backward()
        for i in range(n):
            assert torch.equal(got[3][i], ys2[i].detach()), ("output", n, i)
            if i == skip:
                assert got[1][i] is None and got[2][i] is None
                continue
            assert torch.equal(got[1][i], mods[i].weight.grad), ("dW", n, i)
            assert torch.equal(got[2][i], mods[i].bias.grad), ("db", n, i)
        err = (got[0] - x.grad).norm() / x.grad.norm()
        assert err < 2e-3, ("dx", n, float(err))        # bf16 single-term products, different summation order


def test_colsum_bf16_blocks_matches_fp64(dev):
    """Column sums of a bf16 matrix delivered block by block (the bias gradients of stacked projections: the decoder's memory side,
    the qkv Linear inside the attention node) against fp64, for the 16-byte-load kernel (columns % 8 == 0) and the scalar one, with
    and without accumulation; run twice: bitwise equal (fixed summation order)."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    for R, nblk, blkC, ld in ((8300, 1, 1152, 1152), (8300, 18, 384, 18 * 384), (777, 3, 40, 128), (300, 2, 12, 24), (5, 1, 384, 384)):
        x = torch.randn(R, ld, generator=g).to(dev).to(torch.bfloat16)
        ref = x[:, :nblk * blkC].double().sum(0)
        outs = [torch.full((blkC,), 7.0, device=dev) for _ in range(nblk)]
        K.colsum_bf16_blocks(x, blkC, outs)
        got = torch.cat(outs).double()
        assert (got - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max())) * R ** 0.5, (R, nblk, blkC)
        outs2 = [torch.zeros((blkC,), device=dev) for _ in range(nblk)]
        K.colsum_bf16_blocks(x, blkC, outs2)
        assert torch.equal(torch.cat(outs2), torch.cat(outs)), "not reproducible"


def test_block_with_training_rates_fused_equals_composite(dev):
    """A backbone block with the launch scripts' rates (proj / MLP dropout, DropPath; reference models/cait.py:390-391, 404-416, timm Mlp):
    the fused residual nodes - dropout masks and the per-sample DropPath scale inside the GEMM epilogues (spe_gemm_bf16nt_exd,
    spe_layerscale_residual_bwd16d) - against the composition of the single operators run with the SAME Philox / torch RNG streams:
    identical masks, so outputs and every gradient agree to the rounding order of the two paths."""
    from spe_amd import kernels as K, ops
    from spe_amd.models.cait import LayerScale_Block
    K.set_precision("bf16s")
    torch.manual_seed(21)
    C, H, B, N = 384, 8, 2, 1100                         # 2200 rows: the bf16-copy GEMM path
    blk = LayerScale_Block(C, H, drop=0.1, attn_drop=0.0, drop_path=0.3, init_values=0.5).to(dev).train()
    x0 = torch.randn(B, N, C, device=dev)
    w = torch.randn(B, N, C, device=dev)
    res = {}
    mlp_f16 = K.MLP_F16
    for fused in (True, False, "f16"):
        # (the comparison of the two compositions runs with the MLP on the split operands both take; the third pass is the fused path as
        # shipped - round 5: its MLP forward on single-term fp16 operands - against the split one)
        ops.FUSE_DROP = fused is not False
        K.MLP_F16 = fused == "f16"
        try:
            outs = []
            for trial in range(3):                       # three DropPath draws (kept / dropped samples differ)
                K.manual_seed(77 + trial); torch.manual_seed(5 + trial)
                x = x0.clone().requires_grad_(True)
                for p in blk.parameters():
                    p.grad = None
                y = blk(x)
                (y * w).sum().backward()
                outs.append((y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters()}))
            res[fused] = outs
        finally:
            ops.FUSE_DROP = True
            K.MLP_F16 = mlp_f16
    for (yh, dxh, gh), (yf, dxf, gf) in zip(res["f16"], res[True]):
        assert (yh - yf).norm() <= 3e-4 * yf.norm(), float((yh - yf).norm() / yf.norm())
        assert (dxh - dxf).norm() <= 3e-3 * dxf.norm(), float((dxh - dxf).norm() / dxf.norm())
    for (yf, dxf, gf), (yc, dxc, gc) in zip(res[True], res[False]):
        assert (yf - yc).norm() <= 1e-5 * yc.norm(), float((yf - yc).norm() / yc.norm())
        assert (dxf - dxc).norm() <= 3e-3 * dxc.norm(), float((dxf - dxc).norm() / dxc.norm())
        for n in gc:
            # (proj_l.bias shifts every score of a softmax row alike: its gradient is rounding noise around zero)
            assert (gf[n] - gc[n]).norm() <= 5e-3 * gc[n].norm() + 1e-4, (n, float((gf[n] - gc[n]).norm() / gc[n].norm()))
    # the rates are really applied: the dropped-branch output differs from the deterministic one
    blk.eval()
    with torch.no_grad():
        assert (blk(x0) - res[True][0][0]).norm() > 1e-2 * res[True][0][0].norm()


def test_jitter_pick_kernel_equals_elementwise_composition(dev):
    """One-to-many target jitter (reference models/conditional_detr.py:409-431): the one-launch pick kernel against the elementwise
    torch composition on the SAME uniform draws - identical boxes (the arithmetic is written without contraction), incl. boxes for which
    fewer than ratio - 1 candidates pass the IoU test (tiny boxes: every jitter fails) and images without targets."""
    from spe_amd.models import conditional_detr as cd
    g = torch.Generator().manual_seed(4)
    targets = []
    for m in (7, 0, 23):
        b = torch.rand(m, 4, generator=g) * 0.5 + 0.2
        if m:
            b[0, 2:] = 1e-6                          # degenerate size: IoU of any jittered copy is far below 0.7 or undefined
        targets.append({"boxes": b.to(dev), "labels": torch.randint(0, 20, (m,), generator=g).to(dev), "scores": torch.rand(m, generator=g).to(dev)})
    outs = []
    for kern in (True, False):
        cd.JITTER_KERNEL = kern
        try:
            torch.manual_seed(123)
            outs.append(cd.jitter_targets(targets, 5, 0.1))
        finally:
            cd.JITTER_KERNEL = True
    for a, b in zip(*outs):
        assert a["boxes"].shape == b["boxes"].shape and torch.equal(a["labels"], b["labels"]) and torch.equal(a["scores"], b["scores"])
        assert torch.equal(a["boxes"], b["boxes"])
    assert outs[0][2]["boxes"].shape == (23 * 5, 4) and outs[0][1]["boxes"].shape[0] == 0


def test_deferred_reductions_match_tree_sums_and_handle_shared_parameters(dev):
    """Bias / LayerNorm / LayerScale gradients whose destination is a registered all-reduce bucket are summed by ONE flush launch for
    many producers (spe_reduce_defer_* / csrc/det_reduce.h) instead of by a cross-workgroup tree at the end of every producer: same values
    as the tree up to the order of the fp32 additions, bitwise reproducible, available right after backward() (and autograd.grad()).
    A parameter used by two nodes of the graph (as the decoder's final LayerNorm is) is learned in the first step and keeps out of it.
    Reference: the autograd of nn.LayerNorm / nn.Linear biases under DDP (main.py:172)."""
    import torch.nn as nn
    from spe_amd import kernels as K, ops
    from spe_amd.dp import GradAllReducer
    from spe_amd.models.cait import LayerScale_Block
    from spe_amd.models.layers import LayerNorm
    K.set_precision("bf16s")
    torch.manual_seed(2)
    C, H, B, N = 384, 8, 2, 1100

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = nn.ModuleList([LayerScale_Block(C, H, init_values=0.3) for _ in range(2)])
            self.norm = LayerNorm(C)                       # applied twice: a shared parameter
        def forward(self, x):
            outs = []
            for b in self.blocks:
                x = b(x)
                outs.append(self.norm(x))
            return outs[0] + outs[1]

    net = Net().to(dev).train()
    x = torch.randn(B, N, C, device=dev)
    w = torch.randn(B, N, C, device=dev)
    params = [p for p in net.parameters() if p.requires_grad]
    names = [n for n, p in net.named_parameters() if p.requires_grad]

    def grads(defer, steps):
        old = K.DEFER_REDUCE
        K.DEFER_REDUCE = defer
        try:
            red = GradAllReducer(params, flatten_params=False)
            out = []
            for _ in range(steps):
                red.reset()
                (net(x) * w).sum().backward()
                pending_after_backward = K.lib.load().spe_reduce_pending()      # raw call: the return value is the count
                snap = [p.grad.clone() for p in params]      # read BEFORE finish(): the end-of-backward flush has run
                red.finish()
                out.append((snap, [p.grad.clone() for p in params], pending_after_backward))
            deferring = red._defer
            shared = {n for n, p in zip(names, params) if getattr(p, "_spe_shared", False)}      # (remove() clears the per-parameter marks)
            red.remove()
            return out, deferring, shared
        finally:
            K.DEFER_REDUCE = old

    ref, d0, _ = grads(False, 2)
    got, d1, shared = grads(True, 3)
    assert not d0 and d1
    assert {"norm.weight", "norm.bias"} <= shared
    assert not hasattr(net.norm.weight, "_spe_shared")       # the marks of a removed reducer do not survive into the next one
    for step in (1, 2):                                      # steps with deferral active
        snap, fin, pend = got[step]
        assert pend == 0
        for n, a, b, r in zip(names, snap, fin, ref[1][1]):
            assert torch.equal(a, b), n                      # nothing changed between the end of backward and finish()
            assert (a - r).norm() <= 2e-6 * r.norm() + 1e-7, (n, float((a - r).norm() / (r.norm() + 1e-30)))
    for a, b in zip(got[1][1], got[2][1]):
        assert torch.equal(a, b)                             # same inputs, same weights: bitwise the same sums
