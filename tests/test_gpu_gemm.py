"""GPU parity: jmac_gemm_f32 (small fp32 GEMM of the relation-side projections) in its three forms against float64
matmul, odd shapes and strides, bitwise reproducibility, and the autograd wrapper."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import assert_close


@pytest.mark.parametrize("M,N,K", [(962, 300, 300), (962, 600, 300), (300, 300, 962), (300, 600, 962), (33, 65, 7),
                                   (1, 1, 1), (64, 32, 8), (100, 47, 301), (4806, 300, 300),
                                   (962, 300, 600), (300, 962, 84), (257, 66, 12), (260, 68, 5)])
def test_small_mm_forms(M, N, K):
    from jmac_amd import ops
    gen = torch.Generator().manual_seed(M + N + K)
    A, B = torch.randn(M, K, generator=gen), torch.randn(K, N, generator=gen)
    ref = A.double() @ B.double()
    Ac, Bc = A.cuda(), B.cuda()
    nn_ = ops._gemm(Ac, False, Bc, False, M, N, K)
    assert_close(nn_, ref, 2e-6, what="NN")
    nt = ops._gemm(Ac, False, Bc.t().contiguous(), True, M, N, K)                   # B stored [N,K]
    assert_close(nt, ref, 2e-6, what="NT")
    tn = ops._gemm(Ac.t().contiguous(), True, Bc, False, M, N, K)                   # A stored [K,M]
    assert_close(tn, ref, 2e-6, what="TN")
    tt = ops._gemm(Ac.t().contiguous(), True, Bc.t().contiguous(), True, M, N, K)
    assert_close(tt, ref, 2e-6, what="TT")
    assert torch.equal(nn_, ops._gemm(Ac, False, Bc, False, M, N, K))               # fixed summation order


def test_small_mm_strided_and_autograd():
    from jmac_amd import ops
    gen = torch.Generator().manual_seed(0)
    A = torch.randn(50, 30, generator=gen)
    Wbig = torch.randn(30, 90, generator=gen)
    G = torch.randn(50, 60, generator=gen)
    a64, w64 = A.double().requires_grad_(True), Wbig.double().requires_grad_(True)
    ref = a64 @ w64[:, 30:]                                                         # column slice: row stride 90
    (ref * G.double()).sum().backward()
    ag, wg = A.cuda().requires_grad_(True), Wbig.cuda().requires_grad_(True)
    out = ops.small_mm(ag, wg[:, 30:])
    (out * G.cuda()).sum().backward()
    assert_close(out, ref, 2e-6)
    assert_close(ag.grad, a64.grad, 2e-6)
    assert_close(wg.grad, w64.grad, 2e-6)
