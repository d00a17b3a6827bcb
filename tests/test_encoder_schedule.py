"""Host logic of the encoder node's relation-side schedule (no GPU, no compute): the dependency LEVELS handed to
jmac_gemm_grouped_f32 must be hazard-free -- within one launch no product reads or accumulates into memory another product of
the same launch writes, every input of a product was written in an EARLIER level (or is an operand that existed before), an
accumulating product finds its first writer earlier, and ``balance_levels`` only ever moves deferrable products later."""
import types

import torch

from jmac_amd.encoder import _Chain, _MlpChain, _RelMLP, balance_levels


def _span(t):
    """(start, end) byte range of a tensor's storage footprint (views of one buffer overlap)."""
    if t is None or t.numel() == 0:
        return None
    lo = t.data_ptr()
    hi = lo + ((t.shape[0] - 1) * t.stride(0) + (t.shape[1] - 1) * t.stride(1) + 1) * t.element_size()
    return lo, hi


def _overlap(a, b):
    return a is not None and b is not None and a[0] < b[1] and b[0] < a[1]


def _columns_disjoint(x, y):
    """two views of one row-major buffer that cover different column ranges of the same rows do not overlap element-wise"""
    if x.stride(0) != y.stride(0) or x.stride(0) == x.shape[1]:
        return False
    off = (y.data_ptr() - x.data_ptr()) // x.element_size()
    c = off % x.stride(0)
    return 0 < x.shape[1] <= c or 0 < c + y.shape[1] <= 0 or (c >= x.shape[1] and c + y.shape[1] <= x.stride(0))


def _io(task):
    A, B, Cout, A2, C2, act_src = task._keep
    reads = [A, B, A2, act_src] + ([Cout] if task.accumulate else [])
    return [r for r in reads if r is not None], [w for w in (Cout, C2) if w is not None]


def _check_levels(levels, preexisting):
    written = []                                                  # tensors written by earlier levels
    for li, lv in enumerate(levels):
        outs = []
        for t in lv:
            reads, writes = _io(t)
            for w in writes:
                for o in outs:                                    # two writers of one buffer in one launch
                    assert not _overlap(_span(w), _span(o)) or _columns_disjoint(o, w) or _columns_disjoint(w, o), (li, "write/write")
            outs.extend(writes)
        for t in lv:
            reads, writes = _io(t)
            for r in reads:
                for o in outs:
                    if any(o is w for w in writes) and not t.accumulate:
                        continue
                    if any(o is w for w in writes) and t.accumulate and r is o:
                        continue                                  # its own accumulation target: checked below
                    assert not _overlap(_span(r), _span(o)) or _columns_disjoint(o, r) or _columns_disjoint(r, o), \
                        (li, "a product reads what another product of the same launch writes")
                # every input exists: written earlier, or a pre-existing operand
                ok = any(_overlap(_span(r), _span(p)) for p in preexisting) or any(_overlap(_span(r), _span(w)) for w in written)
                assert ok, (li, "input never written", tuple(r.shape))
            if t.accumulate:
                Cout = t._keep[2]
                assert any(_overlap(_span(Cout), _span(w)) for w in written + preexisting), (li, "accumulate without a first writer")
        written.extend(outs)


def _layer():
    return types.SimpleNamespace(rel_activation="leaky_relu", atv_mlp=types.SimpleNamespace(negative_slope=0.05))


def test_chain_levels_are_hazard_free_forward_and_backward():
    d, nr = 8, 5
    z = lambda *s: torch.zeros(*s)
    R, W1, W2, loop, wc = z(nr, d), z(d, d), z(d, d), z(1, d), z(d, 3 * d)
    ch = _Chain(_layer(), R, W1, W2, loop, wc, d)
    w2g, tt, rr = ch.fwd_tasks()
    _check_levels([[w2g, tt], [rr]], [R, W1, W2, loop, wc])
    dRR, dwc, dR = z(nr + 1, 2 * d), z(d, 3 * d), z(nr, d)
    levels, (dW1, dW2, dloop) = ch.bwd_tasks(dRR, dwc, dR, False)
    pre = [R, W1, W2, loop, wc, dRR, dwc, ch.T, ch.W2g]           # dwc: the node side wrote it first (the chain accumulates)
    _check_levels(levels, pre)
    _check_levels(balance_levels(levels + [[]]), pre)
    assert dW1.shape == W1.shape and dW2.shape == W2.shape and dloop.shape == loop.shape


def test_mlp_chain_levels_are_hazard_free_forward_and_backward():
    d, nr, dh = 8, 5, 12
    z = lambda *s: torch.zeros(*s)
    Ra, L11u, L12u, W1, W2, loop, wc = z(nr, d), z(d, dh), z(dh, d), z(d, d), z(d, d), z(1, d), z(d, 3 * d)
    mc = _MlpChain(_layer(), Ra, L11u, L12u, 0.05, W1, W2, loop, wc, d)
    pre = [Ra, L11u, L12u, W1, W2, loop, wc]
    _check_levels(mc.fwd_tasks(), pre)
    dRR, dwc, dRa = z(nr + 1, 2 * d), z(d, 3 * d), z(nr, d)
    levels, (dW1, dW2, dloop), (dL11u, dL12u) = mc.bwd_tasks(dRR, dwc, dRa, True)
    saved = pre + [dRR, dwc, dRa, mc.M, mc.Wp, mc.W2g, mc.T]      # dRa: conv1_alignment's chain wrote it a level earlier
    _check_levels(levels, saved)
    assert len(levels) == 3 and dL11u.shape == L11u.shape and dL12u.shape == L12u.shape and dW1.shape == W1.shape


def test_relation_mlp_levels_and_balancing():
    d, nr = 8, 5
    z = lambda *s: torch.zeros(*s)
    R, W1, W2 = z(nr, d), z(d, d), z(d, d)
    ml = _RelMLP(R, W1, W2, 0.05)
    _check_levels([[t] for t in ml.fwd_tasks()], [R, W1, W2])
    g, dR = z(nr, d), z(nr, d)
    levels, _ = ml.bwd_tasks(g, dR, False)
    _check_levels(levels, [R, W1, W2, g, ml.M])
    # balancing: a product only ever moves LATER, only if it is deferrable, and nothing is lost or duplicated
    levels3 = [list(levels[0]), list(levels[1]), []]
    before = {id(t): i for i, lv in enumerate(levels3) for t in lv}
    after = balance_levels(levels3)
    pos = {id(t): i for i, lv in enumerate(after) for t in lv}
    assert set(pos) == set(before)
    for lv in levels3:
        for t in lv:
            assert pos[id(t)] >= before[id(t)] and (pos[id(t)] == before[id(t)] or t._defer)


def test_the_checker_sees_a_hazard():
    """(the checks above are only worth something if a wrong schedule fails them)"""
    import pytest
    d, nr = 8, 5
    z = lambda *s: torch.zeros(*s)
    R, W1, W2, loop, wc = z(nr, d), z(d, d), z(d, d), z(1, d), z(d, 3 * d)
    ch = _Chain(_layer(), R, W1, W2, loop, wc, d)
    w2g, tt, rr = ch.fwd_tasks()
    with pytest.raises(AssertionError):
        _check_levels([[w2g, tt, rr]], [R, W1, W2, loop, wc])     # RR = T W2g in the launch that writes T and W2g
    with pytest.raises(AssertionError):
        _check_levels([[rr], [w2g, tt]], [R, W1, W2, loop, wc])   # ... or before it
    levels, _ = ch.bwd_tasks(z(nr + 1, 2 * d), z(d, 3 * d), z(nr, d), False)
    with pytest.raises(AssertionError):
        _check_levels([levels[1], levels[0]], [R, W1, W2, loop, wc, ch.T, ch.W2g])
