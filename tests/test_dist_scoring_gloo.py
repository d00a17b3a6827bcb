"""CPU, world_size 2, gloo: query-sharded scoring (jmac_amd/dist.py: sharded_get_neg, sharded_alignment_test --
BASELINE config 5, alignment-only on 2 GPUs) reproduces the single-process oracle: row split, padded gathers, the
CSLS column-statistic merge and the metric all-reduce.  The rank-local kernels are torch stand-ins INJECTED by this
test (the product's default is the HIP kernels; there is no CPU fallback in jmac_amd)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle.jmac_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Standin:
    @staticmethod
    def sim_topk(a, b, k):
        return orc.topk_lowest_index(a @ b.t(), k)

    @staticmethod
    def sim_matrix(a, b):
        return a @ b.t()

    @staticmethod
    def row_topk_values(s, k):
        return s.topk(k, dim=1).values

    @staticmethod
    def col_topk_values(s, k):
        return s.t().topk(k, dim=1).values

    @staticmethod
    def rank_of_gold(s, gold):
        g = s.gather(1, gold.view(-1, 1))
        idx = torch.arange(s.shape[1]).view(1, -1)
        return ((s > g) | ((s == g) & (idx < gold.view(-1, 1)))).sum(1) + 1


def _case():
    gen = torch.Generator().manual_seed(7)
    n, d = 301, 24                                   # odd: the two ranks get 151 / 150 rows
    e1 = torch.randn(n, d, generator=gen)
    e2 = e1 + 0.8 * torch.randn(n, d, generator=gen)
    ill = torch.randperm(n, generator=gen)[:77]
    return e1, e2, ill


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jmac_amd.dist import shard_rows, sharded_alignment_test, sharded_get_neg
        e1, e2, ill = _case()
        neg = sharded_get_neg(ill.tolist(), e1, e2, 5, kernels=_Standin)
        res = sharded_alignment_test(e1, e2, (1, 5, 10), csls_k=10, kernels=_Standin)
        res0 = sharded_alignment_test(e1, e2, (1, 5, 10), csls_k=0, kernels=_Standin)
        ret[rank] = (neg.numpy(), res, res0, shard_rows(301, world, rank))
    finally:
        dist.destroy_process_group()


def test_sharded_scoring_world2_matches_oracle():
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        out = dict(ret)
    e1, e2, ill = _case()
    want_neg = orc.get_neg(ill.tolist(), e1, e2, 5).numpy()
    want = orc.alignment_test(e1, e2, (1, 5, 10), csls_k=10)
    want0 = orc.alignment_test(e1, e2, (1, 5, 10), csls_k=0)
    assert out[0][3] == (0, 151) and out[1][3] == (151, 301)
    for r in range(world):
        neg, res, res0, _ = out[r]
        assert (neg == want_neg).all()                                   # index work: bit-exact
        for got, ref in ((res, want), (res0, want0)):
            assert got[0] == list(ref[0])
            assert np.allclose(got[1], ref[1], atol=1e-9)                # Hits@k: same ranks
            assert abs(got[2] - ref[2]) < 1e-9 and abs(got[3] - ref[3]) < 1e-12


def test_shard_rows_cover():
    from jmac_amd.dist import shard_rows
    for n in (0, 1, 7, 30000, 10500):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
