"""Run-to-run determinism of the WHOLE training step: forward, both criteria, backward, global-norm clip and
AdamW of the bench.py step, executed twice from the same initial state with foreign kernels scheduled in between, must leave
bitwise equal gradients after the first backward and bitwise equal parameters after two steps.  Every sum across workgroups on
the gradient path is taken in a fixed order (csrc/det_reduce.h); there are no fp32 atomics with more than one addend per address."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("enc_layers,size", [(0, (800, 1333)), (1, (512, 672))])
def test_training_step_is_bitwise_reproducible(dev, enc_layers, size):
    import determinism_step as D
    xs = torch.randn(2048, 1024, device=dev)
    runs = []
    for r in range(3):
        noise = None if r == 0 else (lambda r=r: [(xs @ xs.t()[:, :512 * r]).sum() for _ in range(2 * r)])
        runs.append(D.run_steps(dev, depth=2, H=size[0], W=size[1], batch=2, enc_layers=enc_layers, steps=2, precision="bf16s", noise=noise))
    names, g0, p0, loss0 = runs[0]
    for names_r, g, p, loss in runs[1:]:
        assert loss == loss0
        bad_g = [n for n, a, b in zip(names, g0, g) if a is not None and not torch.equal(a, b)]
        bad_p = [n for n, a, b in zip(names, p0, p) if not torch.equal(a, b)]
        assert not bad_g, ("gradients differ run to run", bad_g[:10], len(bad_g))
        assert not bad_p, ("parameters differ run to run", bad_p[:10], len(bad_p))


def test_column_sums_across_workgroups_are_order_free(dev):
    """The cross-workgroup sums directly: bias-gradient column sums (tall and ragged shapes), LayerNorm backward and the bf16
    conversion with column sums, 30 launches each with foreign work in between - all bitwise equal, and equal to fp64 within fp32."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    xs = torch.randn(2048, 1024, device=dev)
    for (R, C) in ((8300, 384), (8300, 1536), (400, 2048), (1203, 92), (37, 4)):
        x = torch.randn(R, C, generator=g).to(dev)
        ref = None
        for t in range(30):
            if t % 2:
                (xs @ xs.t()[:, :256]).sum()
            out = K.colsum(x, torch.empty(C, device=dev), accumulate=False)
            if ref is None:
                ref = out.clone()
                assert torch.allclose(ref.double(), x.double().sum(0), rtol=1e-5, atol=1e-4 * R ** 0.5)
            assert torch.equal(out, ref), (R, C, t)
    R, C = 8300, 384
    x = torch.randn(R, C, generator=g).to(dev); dy = torch.randn(R, C, generator=g).to(dev); gam = torch.randn(C, generator=g).to(dev)
    y, mean, rstd = K.layernorm_fwd(x, gam, torch.zeros(C, device=dev), 1e-6)[:3]
    ref = None
    for t in range(20):
        if t % 2:
            torch.softmax(xs * (1 + t), dim=1)
        dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
        dx, dg, db = K.layernorm_bwd(dy, x, gam, mean, rstd, dg, db)
        cur = (dx.clone(), dg.clone(), db.clone())
        if ref is None:
            ref = cur
            xh = (x.double() - x.double().mean(1, keepdim=True)) / (x.double().var(1, unbiased=False, keepdim=True) + 1e-6).sqrt()
            assert torch.allclose(dg.double(), (dy.double() * xh).sum(0), rtol=1e-4, atol=1e-2)
            assert torch.allclose(db.double(), dy.double().sum(0), rtol=1e-4, atol=1e-2)
        for a, b in zip(cur, ref):
            assert torch.equal(a, b), t
