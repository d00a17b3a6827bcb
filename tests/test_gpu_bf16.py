"""GPU parity of the bf16-table forms (BASELINE config 3: bf16 union-graph encoder tables + completion scoring).

The kernels widen bf16 to fp32 exactly and accumulate in fp32, so against the oracle evaluated in float64 ON THE SAME
bf16-rounded tables they must agree like the fp32 kernels do (1e-4 relative, north_star's tolerance; measured ~1e-6).
What bf16 costs against the fp32 path is a property of the rounding, stated separately below."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle.jmac_oracle as orc
from util import assert_close, make_args, random_graph, rel_err


def _bf16_tables(n, nr, d, seed):
    gen = torch.Generator().manual_seed(seed)
    PQZ = (torch.randn(n, 3 * d, generator=gen) * 0.3).to(torch.bfloat16)
    RR = (torch.randn(nr, 2 * d, generator=gen) * 0.3).to(torch.bfloat16)
    a = torch.randn(d, generator=gen) * 0.1
    return PQZ, RR, a


@pytest.mark.parametrize("n,nr,e,d,hub,chunk", [(200, 11, 1500, 300, None, None), (500, 30, 6000, 256, 900, 32),
                                                  (64, 5, 300, 64, None, None), (300, 7, 2500, 512, 400, 64),
                                                  (97, 3, 0, 300, None, None), (150, 9, 700, 12, None, None)])
@pytest.mark.parametrize("loop", [True, False])
def test_bf16_aggregate_matches_oracle_on_rounded_tables(n, nr, e, d, hub, chunk, loop):
    from jmac_amd import ops
    from jmac_amd.graph import RelGraph
    rng = np.random.default_rng(n + e)
    ei, et = random_graph(rng, n, nr - 1, e, hub) if e else (np.zeros((2, 0), np.int64), np.zeros(0, np.int64))
    PQZ, RR, a = _bf16_tables(n, nr, d, n)
    g = RelGraph(torch.from_numpy(ei).cuda(), torch.from_numpy(et).cuda(), n, nr, chunk)
    with torch.no_grad():
        out = ops.rel_attn_aggregate(PQZ.cuda(), RR.cuda(), a.cuda(), g, 0.05, nr - 1 if loop else -1, 0.5)
    ref = orc.aggregate_from_tables(PQZ.double(), RR.double(), a.double(), torch.from_numpy(ei), torch.from_numpy(et), 0.05,
                                    nr - 1 if loop else -1, 0.5)
    assert out.dtype == torch.float32
    assert_close(out, ref, 1e-4, what="bf16 aggregate")
    assert rel_err(out, ref) < 2e-5
    # the layout the layers produce: halves padded to a multiple of 8 elements (d = 300 -> 304, zero pad columns) -- the
    # half-wave kernel's 16-byte lane loads; same values, so the same result up to the summation order
    if ops.bf16_pad(d) != d:
        with torch.no_grad():
            out_p = ops.rel_attn_aggregate(ops.pad_table(PQZ, d, 3).cuda(), ops.pad_table(RR, d, 2).cuda(), a.cuda(), g, 0.05,
                                           nr - 1 if loop else -1, 0.5)
        assert out_p.shape == out.shape
        assert_close(out_p, ref, 1e-4, what="bf16 aggregate, padded halves")
        assert rel_err(out_p, ref) < 2e-5


def test_bf16_tables_refuse_autograd():
    from jmac_amd import ops
    from jmac_amd.graph import RelGraph
    PQZ, RR, a = _bf16_tables(10, 3, 8, 0)
    g = RelGraph(torch.tensor([[0, 1], [2, 3]]).cuda(), torch.tensor([0, 1]).cuda(), 10, 3)
    with pytest.raises(RuntimeError):
        ops.rel_attn_aggregate(PQZ.cuda().requires_grad_(True), RR.cuda(), a.cuda(), g, 0.05, 2, 0.5)


@pytest.mark.parametrize("B,N,d", [(37, 301, 48), (128, 1000, 300), (5, 64, 7), (1000, 2111, 256)])
def test_bf16_l1_scores(B, N, d):
    from jmac_amd import scoring
    gen = torch.Generator().manual_seed(B + N)
    er, tab = torch.randn(B, d, generator=gen).to(torch.bfloat16), torch.randn(N, d, generator=gen).to(torch.bfloat16)
    ref = torch.cdist(er.double(), tab.double(), p=1)
    out = scoring.l1_scores(er.cuda(), tab.cuda())
    assert out.dtype == torch.float32
    assert_close(out, ref, 1e-5)
    out2 = scoring.l1_scores(er.cuda(), tab.cuda(), out=out.clone(), accumulate=True)
    assert_close(out2, 2 * ref, 1e-5)


def test_bf16_layer_mode_tracks_fp32_layer():
    """Layer in table_dtype=bfloat16 against the same layer in fp32: the difference is the bf16 rounding of the
    projected tables (2^-9 relative per entry), amplified by BatchNorm's 1/std; bounded here at 5e-2 of the tanh range."""
    from jmac_amd.layer import RelationAwareLayer
    rng = np.random.default_rng(5)
    n, nr, d, e = 3000, 40, 300, 9000
    ei, et = random_graph(rng, n, nr, e)
    torch.manual_seed(1)
    lay = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args()).cuda().eval()
    X, R = (torch.randn(n, d) * 0.3).cuda(), (torch.randn(nr, d) * 0.3).cuda()
    ei_t, et_t = torch.from_numpy(ei).cuda(), torch.from_numpy(et).cuda()
    with torch.no_grad():
        ref = lay(X, R, ei_t, et_t)
        lay.table_dtype = torch.bfloat16
        out = lay(X, R, ei_t, et_t)
    assert out.dtype == torch.float32
    assert (out - ref).abs().max().item() < 5e-2
    assert (out - ref).abs().mean().item() < 5e-3
    lay.train()
    with pytest.raises(RuntimeError):
        lay(X.requires_grad_(True), R, ei_t, et_t)
