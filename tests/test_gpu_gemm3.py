"""GPU: jmac_gemm_nt_x3_f32 (fp32 GEMM on the bf16 matrix cores: three-term split, six products) against float64 --
per product its error is at the level of an fp32 GEMM's, for ragged shapes, rows of very different magnitude, every tile
shape, and through autograd (ops.mm_x3).  The op is experimental and NOT on the layer's path: the bf16 MFMA accumulates
with a small negative bias that the last test measures (and that costs the full-size step its 1e-4 gradient parity)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(11805, 900, 300), (11805, 300, 900), (4097, 300, 600), (2500, 33, 52), (129, 257, 8),
                                   (64, 64, 32), (5000, 928, 300), (1, 5, 4)])
def test_gemm_nt_x3_matches_float64(M, N, K):
    from jmac_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.exp(torch.rand(M, 1, generator=g) * 8 - 4)     # row scales e^-4 .. e^4
    B = torch.randn(N, K, generator=g) * 0.05
    # exact-integer check of the operand / accumulator maps with an asymmetric B (small integers: every term is exact)
    Ai = torch.randint(-3, 4, (M, K), generator=g).float()
    Bi = (torch.arange(N).view(-1, 1) % 5 + torch.arange(K).view(1, -1) % 3).float()
    assert torch.equal(ops.gemm_nt_x3(Ai.cuda(), Bi.cuda()).cpu(), Ai @ Bi.t())
    C = ops.gemm_nt_x3(A.cuda(), B.cuda()).cpu().double()
    ref = A.double() @ B.double().t()
    scale = A.double().abs() @ B.double().abs().t()
    err = ((C - ref).abs() / scale.clamp_min(1e-300)).max().item()
    err32 = (((A @ B.t()).double() - ref).abs() / scale.clamp_min(1e-300)).max().item()     # a CPU fp32 GEMM, for scale
    assert err <= max(2.0 * err32, 6e-7), (err, err32)


def test_mm_x3_autograd_matches_torch():
    from jmac_amd import ops
    g = torch.Generator().manual_seed(0)
    A = (torch.randn(3000, 600, generator=g) * 0.3).cuda().requires_grad_(True)
    W = (torch.randn(600, 300, generator=g) * 0.05).cuda().requires_grad_(True)
    G = torch.randn(3000, 300, generator=g).cuda()
    out = ops.mm_x3(A, W)
    out.backward(G)
    ga, gw = A.grad.clone(), W.grad.clone()
    A.grad = W.grad = None
    ref = torch.mm(A.double(), W.double())
    ref.backward(G.double())
    for got, want in ((out, ref), (ga, A.grad), (gw, W.grad)):
        assert (got.double() - want).abs().max().item() <= 2e-6 * want.abs().max().item()


def test_bf16_mfma_accumulation_bias_is_measurable():
    """Why the op is not wired in: on all-positive operands the mean SIGNED error of the split-bf16 product is ~1e-8 of the
    result, two orders above the fp32 GEMM's, although its rms error is lower -- a bias, not noise."""
    from jmac_amd import ops
    g = torch.Generator().manual_seed(1)
    A = (torch.rand(4096, 912, generator=g) + 0.5).cuda()
    B = (torch.rand(512, 912, generator=g) + 0.5).cuda()
    ref = A.double() @ B.double().t()
    e3 = ((ops.gemm_nt_x3(A, B).double() - ref) / ref)
    e32 = ((torch.mm(A, B.t()).double() - ref) / ref)
    assert e3.pow(2).mean().sqrt().item() < 1e-6 and e3.abs().max().item() < 5e-6           # accurate per element ...
    assert abs(e3.mean().item()) > 5 * abs(e32.mean().item())                               # ... but biased
    assert abs(e3.mean().item()) < 1e-7
