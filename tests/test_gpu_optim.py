"""GPU parity: jmac_amd.optim.Adam / AdamW (jmac_adam_step_f32, one launch over all parameter tensors) against torch.optim.Adam --
the optimizer the reference steps once per batch (train.py:358-359, built at :406-407 with the defaults) -- run in float64 on the
CPU on the same gradients, and held to the error torch's own fp32 implementation makes; per-parameter step counts, state_dict
exchange with torch.optim.Adam in both directions, hipGraph replay, more tensors than one launch takes."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1,), (3,), (5, 1), (300,), (4099,), (300, 300), (961, 300), (2000, 300), (4096,), (8192,), (17, 13, 3)]


def _params(seed, shapes=SHAPES, dtype=torch.float64, device="cpu"):
    gen = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(*s, generator=gen) * 0.3).to(device=device, dtype=dtype)) for s in shapes]


def _grads(seed, shapes, step):
    gen = torch.Generator().manual_seed(seed * 1000 + step)
    return [torch.randn(*s, generator=gen) * (0.05 + 0.01 * step) for s in shapes]


def _run(opt, ps, seed, steps, shapes=SHAPES, skip=None, first=0):
    for k in range(first, first + steps):
        for i, (p, g) in enumerate(zip(ps, _grads(seed, shapes, k))):
            p.grad = None if (skip is not None and skip(i, k)) else g.to(device=p.device, dtype=p.dtype)
        opt.step()


def _close(ours, ref64, torch32, p0, what):
    """ours within 4x (+ one fp32 ulp of the parameter) of the error torch's fp32 Adam makes against float64."""
    for i, (a, r, b, z) in enumerate(zip(ours, ref64, torch32, p0)):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        scale = (r.detach() - z).abs().max().item() + 1e-30
        e_ours = (a - r.detach()).abs().max().item()
        e_torch = (b - r.detach()).abs().max().item()
        ulp = r.detach().abs().max().item() * 2.0 ** -23
        assert e_ours <= 4 * e_torch + 2 * ulp + 1e-7 * scale, (what, i, e_ours, e_torch, ulp, scale)


@pytest.mark.parametrize("kw", [dict(), dict(lr=3e-3, betas=(0.8, 0.99), eps=1e-6), dict(weight_decay=0.01), dict(maximize=True),
                                dict(weight_decay=0.05, maximize=True)])
def test_adam_matches_torch(kw):
    from jmac_amd import optim
    p64, p32, pj = _params(1), _params(1, dtype=torch.float32, device="cuda"), _params(1, dtype=torch.float32, device="cuda")
    p0 = [p.detach().clone() for p in p64]
    o64, o32, oj = torch.optim.Adam(p64, **kw), torch.optim.Adam(p32, **kw), optim.Adam(pj, **kw)
    for o, ps in ((o64, p64), (o32, p32), (oj, pj)):
        _run(o, ps, 7, 25)
    _close(pj, p64, p32, p0, "param")
    for k in ("exp_avg", "exp_avg_sq"):
        _close([oj.state[p][k] for p in pj], [o64.state[p][k] for p in p64], [o32.state[p][k] for p in p32],
               [torch.zeros_like(p) for p in p64], k)
    assert all(float(oj.state[p]["step"]) == 25.0 for p in pj)
    assert len({id(oj.state[p]["step"]) for p in pj}) == 1                 # one shared device count: one launch per step


def test_adamw_matches_torch():
    from jmac_amd import optim
    p64, p32, pj = _params(2), _params(2, dtype=torch.float32, device="cuda"), _params(2, dtype=torch.float32, device="cuda")
    p0 = [p.detach().clone() for p in p64]
    kw = dict(lr=2e-3, weight_decay=0.02)
    o64, o32, oj = torch.optim.AdamW(p64, **kw), torch.optim.AdamW(p32, **kw), optim.AdamW(pj, **kw)
    for o, ps in ((o64, p64), (o32, p32), (oj, pj)):
        _run(o, ps, 9, 20)
    _close(pj, p64, p32, p0, "param")


def test_step_counts_are_per_parameter():
    """A parameter without a gradient is skipped and its count stands still (the reference steps two optimizers over all parameters,
    each loss reaching a subset): tensors 0-3 step every time, 4-7 from call 3 on, 8-10 on even calls only."""
    from jmac_amd import optim
    skip = lambda i, k: (4 <= i < 8 and k < 3) or (i >= 8 and k % 2 == 1)
    p64, p32, pj = _params(3), _params(3, dtype=torch.float32, device="cuda"), _params(3, dtype=torch.float32, device="cuda")
    p0 = [p.detach().clone() for p in p64]
    o64, o32, oj = torch.optim.Adam(p64), torch.optim.Adam(p32), optim.Adam(pj)
    for o, ps in ((o64, p64), (o32, p32), (oj, pj)):
        _run(o, ps, 11, 12, skip=skip)
    _close(pj, p64, p32, p0, "param")
    assert [float(oj.state[p]["step"]) for p in pj] == [float(o64.state[p]["step"]) for p in p64]
    assert len({id(oj.state[p]["step"]) for p in pj}) == 3


@pytest.mark.parametrize("direction", ["torch->jmac", "jmac->torch"])
def test_state_dict_exchange_with_torch(direction):
    from jmac_amd import optim
    p64 = _params(4)
    p0 = [p.detach().clone() for p in p64]
    o64 = torch.optim.Adam(p64, lr=2e-3)
    _run(o64, p64, 13, 16)
    pa = _params(4, dtype=torch.float32, device="cuda")
    first, second = (torch.optim.Adam, optim.Adam) if direction == "torch->jmac" else (optim.Adam, torch.optim.Adam)
    oa = first(pa, lr=2e-3)
    _run(oa, pa, 13, 8)
    sd = copy.deepcopy(oa.state_dict())
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    ob = second(pa, lr=2e-3)
    ob.load_state_dict(sd)
    _run(ob, pa, 13, 8, first=8)
    p32 = _params(4, dtype=torch.float32, device="cuda")
    o32 = torch.optim.Adam(p32, lr=2e-3)
    _run(o32, p32, 13, 16)
    _close(pa, p64, p32, p0, "param")
    assert all(float(ob.state[p]["step"]) == 16.0 for p in pa)


def test_graph_replay_advances_the_device_step_count():
    from jmac_amd import optim
    shapes = SHAPES
    pj = _params(5, dtype=torch.float32, device="cuda")
    p64 = _params(5)
    p0 = [p.detach().clone() for p in p64]
    oj, o64 = optim.Adam(pj, lr=1e-3), torch.optim.Adam(p64, lr=1e-3)
    static = [torch.zeros_like(p) for p in pj]
    for p, g in zip(pj, static):
        p.grad = g

    def feed(k):
        for s, g in zip(static, _grads(17, shapes, k)):
            s.copy_(g)

    feed(0)
    oj.step()                                     # eager first step: state + the step count are created here
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    feed(1)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        with torch.cuda.graph(graph, stream=st):
            oj.step()
    torch.cuda.synchronize()
    # the capture itself does not run the kernel: replay for steps 1 .. 6
    for k in range(1, 7):
        feed(k)
        graph.replay()
    torch.cuda.synchronize()
    _run(o64, p64, 17, 7)
    p32 = _params(5, dtype=torch.float32, device="cuda")
    o32 = torch.optim.Adam(p32, lr=1e-3)
    _run(o32, p32, 17, 7)
    _close(pj, p64, p32, p0, "param")
    assert float(oj.state[pj[0]]["step"]) == 7.0


def test_more_tensors_than_one_launch():
    from jmac_amd import optim
    shapes = [(37 + i,) for i in range(150)] + [(300, 300)]
    p64, p32, pj = (_params(6, shapes), _params(6, shapes, dtype=torch.float32, device="cuda"),
                    _params(6, shapes, dtype=torch.float32, device="cuda"))
    p0 = [p.detach().clone() for p in p64]
    o64, o32, oj = torch.optim.Adam(p64), torch.optim.Adam(p32), optim.Adam(pj)
    for o, ps in ((o64, p64), (o32, p32), (oj, pj)):
        _run(o, ps, 19, 5, shapes=shapes)
    _close(pj, p64, p32, p0, "param")
    assert float(oj.state[pj[0]]["step"]) == 5.0            # three launches per step, the count advanced once


def test_rejects_what_it_does_not_do():
    from jmac_amd import optim
    from jmac_amd._lib import AdamTask, lib
    p = torch.nn.Parameter(torch.randn(8))
    with pytest.raises(NotImplementedError):
        optim.Adam([p], amsgrad=True)
    with pytest.raises(ValueError):
        optim.Adam([p], betas=(1.0, 0.999))
    o = optim.Adam([p])
    p.grad = torch.randn(8)
    with pytest.raises(TypeError):
        o.step()                                            # a CPU parameter: there is no CPU path
    step, aux = torch.zeros((), device="cuda"), torch.tensor([1.0, 1.0, 0.0], dtype=torch.float64, device="cuda")
    tasks = (AdamTask * 1)()
    assert lib().jmac_adam_step_f32(tasks, 1, step.data_ptr(), aux.data_ptr(), 1e-3, 0.9, 1.5, 1e-8, 0.0, 0, 0, None) != 0
    assert lib().jmac_adam_step_f32(tasks, 1, None, aux.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 0, None) != 0
    assert lib().jmac_adam_step_f32(tasks, 1, step.data_ptr(), None, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 0, None) != 0
    # no tensors at all: the count and the powers still advance
    assert lib().jmac_adam_step_f32(tasks, 0, step.data_ptr(), aux.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 0, None) == 0
    torch.cuda.synchronize()
    assert float(step) == 1.0 and aux.tolist() == [0.9, 0.999, 0.0]


def test_changed_betas_restart_the_powers():
    from jmac_amd import optim
    p64, p32, pj = _params(8), _params(8, dtype=torch.float32, device="cuda"), _params(8, dtype=torch.float32, device="cuda")
    p0 = [p.detach().clone() for p in p64]
    o64, o32, oj = torch.optim.Adam(p64), torch.optim.Adam(p32), optim.Adam(pj)
    for o, ps in ((o64, p64), (o32, p32), (oj, pj)):
        _run(o, ps, 23, 6)
        o.param_groups[0]["betas"] = (0.85, 0.98)
        o.param_groups[0]["lr"] = 5e-4
        _run(o, ps, 23, 6, first=6)
    _close(pj, p64, p32, p0, "param")


def test_param_groups_empty_tensors_and_strided_gradients():
    """Two groups with their own hyper-parameters (each has its own device step count), a zero-element parameter, and gradients that
    arrive as expanded / transposed views (autograd hands those out: sum().backward(), .t())."""
    from jmac_amd import optim
    shapes = [(300, 40), (40,), (0, 7), (64, 64), (5,)]

    def build(dtype, device, cls):
        ps = _params(10, shapes, dtype=dtype, device=device)
        return ps, cls([{"params": ps[:3], "lr": 2e-3, "weight_decay": 0.01}, {"params": ps[3:], "betas": (0.7, 0.95)}], lr=1e-3)

    (p64, o64), (p32, o32), (pj, oj) = (build(torch.float64, "cpu", torch.optim.Adam), build(torch.float32, "cuda", torch.optim.Adam),
                                        build(torch.float32, "cuda", optim.Adam))
    p0 = [p.detach().clone() for p in p64]
    for k in range(8):
        for ps in (p64, p32, pj):
            gs = _grads(29, shapes, k)
            for i, (p, g) in enumerate(zip(ps, gs)):
                g = g.to(device=p.device, dtype=p.dtype)
                if i == 1:
                    g = g[:1].expand(40)                       # stride 0
                if i == 3:
                    g = g.t()                                  # transposed view
                p.grad = g
        for o in (o64, o32, oj):
            o.step()
    _close([p for i, p in enumerate(pj) if i != 2], [p for i, p in enumerate(p64) if i != 2], [p for i, p in enumerate(p32) if i != 2],
           [p for i, p in enumerate(p0) if i != 2], "param")
    assert len({id(oj.state[p]["step"]) for p in pj}) == 2
