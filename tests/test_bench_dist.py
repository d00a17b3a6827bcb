"""CPU: the strong-scaled sharded workload of bench.py (--scaling strong: north_star's 8-GPU configuration) and the scaling model
it puts on the line."""
import numpy as np


def test_strong_graph_is_the_same_graph_at_every_world_size():
    """One global graph, fixed row blocks: the ranks of a world of 2, 4 or 8 generate exactly the edges the single rank of a
    world of 1 generates, each rank a contiguous destination range with an equal share of the edges."""
    import bench_dist
    n, e, nr = 8000, 160000, 50
    ei1, et1, n1, e1 = bench_dist._strong_graph(0, 1, n, e, nr)
    assert (n1, e1) == (n, e) and ei1.shape == (2, e) and ei1[0].max() < n and ei1[1].max() < n
    for W in (2, 4, 8):
        parts = [bench_dist._strong_graph(r, W, n, e, nr) for r in range(W)]
        assert all(p[2] == n // W and p[3] == e // W for p in parts)
        for r, p in enumerate(parts):
            assert p[0][0].min() >= r * (n // W) and p[0][0].max() < (r + 1) * (n // W)      # own destination range
        assert (np.concatenate([p[0] for p in parts], axis=1) == ei1).all()
        assert (np.concatenate([p[1] for p in parts]) == et1).all()


def test_scaling_model_is_consistent():
    """The model reproduces the measured step at the measured world size, counts edges per step like the line's value, and
    its exchange term follows the per-link slab (N / W rows x 2d x bytes over one 153 GB/s link at 0.8)."""
    import bench_dist
    m = bench_dist.scaling_model(110.0, 10.5, 20.0, 2, 1_000_000, 20_000_000, 300, 1, False)
    p1, p8 = m["predicted"]["1"], m["predicted"]["8"]
    assert abs(p1["step_ms"] - 110.0) < 1e-9 and p1["exchange_ms"] == 0.0
    assert abs(p1["edges_per_s"] - 2 * 20_000_000 / 0.110) < 1e-3
    slab_ms = 1_000_000 * 600 * 4 / (153e9 * 0.8) * 1e3
    assert abs(p8["exchange_ms"] - 2 * 2 * slab_ms) < 1e-9 and abs(p8["compute_ms"] - 110.0) < 1e-9       # weak: per-GPU work fixed
    s = bench_dist.scaling_model(110.0, 10.5, 20.0, 2, 2_000_000, 200_000_000, 300, 1, True)
    q8 = s["predicted"]["8"]
    assert abs(q8["compute_ms"] - 110.0 / 8) < 1e-9                                                      # strong: 1/8 of the rows and edges
    assert abs(q8["exchange_ms"] - 2 * 2 * (250_000 * 600 * 4 / (153e9 * 0.8) * 1e3)) < 1e-9
    assert s["predicted"]["2"]["speedup_vs_1"] < 2 and q8["speedup_vs_1"] < 8
    # the pipelined figure hides min(forward aggregation, all-gather x 3/4) per layer and pays the merge pass; nothing at one rank
    assert abs(p1["pipelined_step_ms"] - p1["step_ms"]) < 1e-9 and p8["pipeline_chunks"] == 4
    over_ms = 7.5                                                   # measured cost per layer at 1M rows / 20M edges per GPU
    assert abs((p8["step_ms"] - p8["pipelined_step_ms"]) - 2 * (min(10.5, slab_ms * 0.75) - over_ms)) < 1e-9
    assert p8["pipelined_step_ms"] < p8["step_ms"]


def test_scaling_model_takes_the_rehearsed_aggregation_where_it_was_measured():
    """Round 6: the aggregation kernels were timed on ONE GPU on the shapes a rank has at world 2 / 4 / 8 (its 1M rows / 20M edges
    gathering from the W x 1M-row table; profiles/r6_rehearse_w*.json).  The weak-scaling model uses those figures instead of the
    world-1 rate it used to carry over, keeps the carried-over prediction beside them, and leaves the strong models alone."""
    import json
    import os
    import bench_dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    reh = {W: json.load(open(os.path.join(root, "profiles", "r6_rehearse_w%d.json" % W))) for W in (2, 4, 8)}
    for W, r in reh.items():
        assert r["world"] == W and r["table_rows"] == W * 1_000_000 and r["checks"]["ok"] is True
        assert r["checks"]["phased_equals_one_call_bitwise"] is True
    assert reh[8]["table_GB"] > 2 ** 32 / 1e9                                   # the table is past 4 GiB: 64-bit row offsets
    step, f1, b1 = 108.0, 10.9, 20.2
    old = bench_dist.scaling_model(step, f1, b1, 2, 1_000_000, 20_000_000, 300, 1, False)
    new = bench_dist.scaling_model(step, f1, b1, 2, 1_000_000, 20_000_000, 300, 1, False, rehearsal=reh)
    assert new["carried_over"]["predicted"] == old["predicted"] and new["carried_over"]["band"] == old["band"]
    other = step - 2 * (f1 + b1)
    for W in (2, 4, 8):
        r = reh[W]
        want = 2 * (r["one_piece_fwd_ms"] + r["one_call_bwd_ms"]) + other
        assert abs(new["predicted"][str(W)]["compute_ms"] - want) < 1e-9
        assert new["predicted"][str(W)]["exchange_ms"] == old["predicted"][str(W)]["exchange_ms"]
        cmp_ = new["carried_over_vs_rehearsed_speedup"][str(W)]
        assert cmp_["carried_over"] == old["predicted"][str(W)]["speedup_vs_1"]
        assert cmp_["rehearsed"] == new["predicted"][str(W)]["speedup_vs_1"]
    # the 19.2 GB table costs the backward more than the forward: the rehearsed 8-rank figure sits below the carried-over one
    assert new["predicted"]["8"]["speedup_vs_1"] < old["predicted"]["8"]["speedup_vs_1"]
    assert new["predicted"]["1"] == old["predicted"]["1"]
    # a strong-scaled model is a different per-rank shape: the rehearsal is not applied
    s_old = bench_dist.scaling_model(step, f1, b1, 2, 2_000_000, 200_000_000, 300, 1, True, run_rows=1_000_000, run_edges=20_000_000)
    s_new = bench_dist.scaling_model(step, f1, b1, 2, 2_000_000, 200_000_000, 300, 1, True, run_rows=1_000_000, run_edges=20_000_000,
                                     rehearsal=reh)
    assert s_new["predicted"] == s_old["predicted"] and "carried_over" not in s_new


def test_strong_model_takes_the_strong_rehearsal_and_only_that():
    """rehearse_strong(): the aggregation of rank 0's share of the 2M / 200M graph at world 1, 2, 4, 8, measured on one GPU.  The
    strong model's compute term becomes those figures + the run's non-aggregation time per row; a weak rehearsal is ignored by a
    strong model and a strong one by a weak model."""
    import bench_dist
    strong = {W: {"fwd_ms": 120.0 / W + 1.0, "bwd_ms": 230.0 / W + 2.0, "local_rows": 2_000_000 // W, "local_edges": 200_000_000 // W}
              for W in (1, 2, 4, 8)}
    step, f1, b1 = 108.0, 10.9, 20.2
    kw = dict(run_rows=1_000_000, run_edges=20_000_000)
    old = bench_dist.scaling_model(step, f1, b1, 2, 2_000_000, 200_000_000, 300, 1, True, **kw)
    new = bench_dist.scaling_model(step, f1, b1, 2, 2_000_000, 200_000_000, 300, 1, True, rehearsal=strong, **kw)
    other = step - 2 * (f1 + b1)
    for W in (1, 2, 4, 8):
        want = 2 * (strong[W]["fwd_ms"] + strong[W]["bwd_ms"]) + other * (2_000_000 / W) / 1_000_000
        assert abs(new["predicted"][str(W)]["compute_ms"] - want) < 1e-9
        assert new["predicted"][str(W)]["exchange_ms"] == old["predicted"][str(W)]["exchange_ms"]
    assert new["carried_over"]["predicted"] == old["predicted"]
    weak_model = bench_dist.scaling_model(step, f1, b1, 2, 1_000_000, 20_000_000, 300, 1, False, rehearsal=strong)
    assert "carried_over" not in weak_model
    assert weak_model["predicted"] == bench_dist.scaling_model(step, f1, b1, 2, 1_000_000, 20_000_000, 300, 1, False)["predicted"]
