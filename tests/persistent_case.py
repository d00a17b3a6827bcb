"""Shared by tests/test_gpu_persistent_oracle.py and its subprocess worker: ONE seeded graph large enough for the
PERSISTENT-GRID forms of the aggregation kernels (more than 65 536 by-destination items without inline entries:
jmac_amd/graph.py INLINE_EDGES_MAX_ITEMS, aggregate.hip launch_rel_attn_fwd), seeded tables on it, and the oracle
(oracle/jmac_oracle.py aggregate_from_tables_sliced -- the reference's per-destination softmax / weighted sum of
modules/helper/message_passing.py:24-28 on the node / relation tables of src/jmac_model.py:75-88)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_ENT, N_EDGE, N_REL = 80000, 600000, 100
SLOPE, OUT_SCALE = 0.05, 0.5
HUB_SOURCES = ((5, 2000), (N_ENT // 2, 700), (N_ENT - 3, 1300))      # sources with enough out-edges to be split in pass B


def graph(n=N_ENT, e=N_EDGE, nr=N_REL, seed=501):
    """Power-law destinations (hubs of 10^4 in-edges), Zipf relation types, uniform sources + three hub sources.
    -> (edge_index [2,E] int64, edge_type [E] int64, N, nrel) with nrel = nr + 1 (the loop row)."""
    from jmac_amd import synth
    ei, et, n, nrel = synth.power_law_graph(n, e, nr, seed=seed)
    rng = np.random.default_rng(seed + 1)
    pos, off = rng.permutation(e), 0
    for hub, cnt in HUB_SOURCES:
        cnt = min(cnt, e // 8)
        ei[1, pos[off:off + cnt]] = min(hub, n - 1)
        off += cnt
    return ei, et, n, nrel


def tables(n, nrel, d, seed=7):
    """fp32 [P|Q|Z] [n,3d], [Rq|Rz] [nrel,2d], a [d], upstream gradient G [n,d] (CPU generator: the same bits in every process)."""
    gen = torch.Generator().manual_seed(seed)
    PQZ = torch.randn(n, 3 * d, generator=gen) * 0.3
    RR = torch.randn(nrel, 2 * d, generator=gen) * 0.3
    a = torch.randn(d, generator=gen) * 0.1
    G = torch.randn(n, d, generator=gen)
    return PQZ, RR, a, G


def oracle_aggregate(PQZ, RR, a, ei, et, loop_rel, dtype, G=None, kink_mask=None, nslices=8):
    """out_scale * (nb + Z - Rz[loop]) of the oracle in ``dtype`` (+ (dPQZ, dRR, da) of sum(out * G) with G), evaluated in
    destination slices (oracle.aggregate_from_tables_sliced)."""
    import oracle.jmac_oracle as orc
    return orc.aggregate_from_tables_sliced(PQZ, RR, a, torch.from_numpy(ei), torch.from_numpy(et), SLOPE, loop_rel, OUT_SCALE,
                                            dtype, G, kink_mask, nslices)


def rel_err(got, ref):
    got = torch.as_tensor(got).detach().double().cpu()
    ref = torch.as_tensor(ref).detach().double().cpu()
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
