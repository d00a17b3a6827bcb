"""GPU: the fused encoder nodes (jmac_amd/encoder.py: JMAC.forward_name / forward_no_name as one autograd node each) and
the kernels they add -- jmac_gemm_grouped_f32, jmac_bn_tanh_{fwd,bwd}2_f32, jmac_row_normalize_drop_{fwd,bwd}_f32.

The reference fixtures reach the fused node through every model-level test (test_gpu_model.py, test_gpu_e2e.py,
test_gpu_ja_oracle.py, ...); here it is additionally held to the op-by-op path of jmac_amd.model -- an independent second
implementation of the same function (src/jmac_model.py:172-220) -- on outputs and on every gradient, for each pattern of
used outputs a caller produces (alignment loss: align_out only; completion loss: completion layers + relation layers only;
the bench step: all of them)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import assert_close, random_graph

DEV = "cuda"


def _args(d, no_name=False, dropout=0.0):
    return types.SimpleNamespace(dim=d, dropout=dropout, leaky_relu_w=0.05, comp_op="sub", num_gcn_layer=2, num_negative=5,
                                 margin_align=1.0, margin_completion=5.0, batch_size=64, no_name_info=no_name, device=DEV)


def _model(d, n, nr, di, no_name, seed):
    from jmac_amd.model import JMAC
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    m = JMAC(_args(d, no_name), rng.standard_normal((n, di)).astype(np.float32), nr, n).to(DEV)
    m.ent_info_att = m.ent_info_att.to(DEV)
    with torch.no_grad():
        for lay in (m.conv1_alignment, m.conv2_alignment, m.conv1_completion):   # non-trivial BN affine
            lay.bn.weight.add_(0.1 * torch.randn_like(lay.bn.weight))
            lay.bn.bias.add_(0.1 * torch.randn_like(lay.bn.bias))
    return m


def _run(m, fused, ei, et, n, nr, use, G):
    """One forward + backward; ``use`` selects which outputs feed the scalar loss."""
    m.fused_encoder = fused
    for lay in (m.conv1_alignment, m.conv2_alignment, m.conv1_completion):
        lay.fused = fused                                         # op by op all the way down: no node of encoder.py involved
    m.zero_grad(set_to_none=True)
    state = {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
    align_out, comp, rel = m.forward_base(ei, et, [0, n], [0, nr])
    loss = 0
    if "align" in use:
        loss = loss + (align_out * G["align"]).sum()
    if "comp" in use:
        loss = loss + (comp[1] * G["c1"]).sum() + (comp[0] * G["c0"]).sum()
    if "rel" in use:
        loss = loss + (rel[1] * G["r1"]).sum() + (rel[0] * G["r0"]).sum()
    loss.backward()
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}
    after = {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
    m.load_state_dict(state, strict=False)                       # both runs start from the same BN statistics
    return (align_out.detach(), comp[1].detach(), rel[1].detach()), grads, after


@pytest.mark.parametrize("use", [("align", "comp", "rel"), ("align",), ("comp", "rel"), ("rel",), ("comp",)])
@pytest.mark.parametrize("no_name", [False, True], ids=["name", "no-name"])
def test_fused_encoder_equals_op_by_op_path(no_name, use):
    if no_name and use == ("align",):
        use = ("comp",)                                          # forward_no_name's "align" output IS completion layer 1
    n, nr, d, di = 700, 37, 32, 20
    rng = np.random.default_rng(5)
    ei, et = random_graph(rng, n, nr, 2600, hub=300)
    ei, et = torch.from_numpy(ei).to(DEV), torch.from_numpy(et).to(DEV)
    m = _model(d, n, nr, di, no_name, 11)
    from jmac_amd import encoder
    assert encoder.supported(m, None if no_name else di)
    gen = torch.Generator(device=DEV).manual_seed(3)
    G = {k: torch.randn(s, device=DEV, generator=gen) for k, s in
         (("align", (n, d)), ("c1", (n, d)), ("c0", (n, d)), ("r1", (nr, d)), ("r0", (nr, d)))}
    m.train()
    out_f, g_f, bn_f = _run(m, True, ei, et, n, nr, use, G)
    out_o, g_o, bn_o = _run(m, False, ei, et, n, nr, use, G)
    for a, b, what in zip(out_f, out_o, ("align_out", "c1", "rel_c1")):
        assert_close(a, b, 2e-5, 1e-6, what)
    for k in bn_o:
        assert_close(bn_f[k], bn_o[k], 1e-5, 1e-7, k)
    scale = max(float(g.abs().max()) for g in g_o.values() if g is not None)
    n_checked = 0
    for k, ref in g_o.items():
        got = g_f[k]
        if ref is None:
            assert got is None or float(got.abs().max()) == 0.0, k
            continue
        assert got is not None, k
        atol = 1e-4 * scale if k.endswith("loop_rel") else 1e-6 * scale       # loop_rel: zero gradient under train-mode BN
        assert_close(got, ref, 1e-4, atol, "grad " + k)
        n_checked += 1
    assert n_checked >= 3, n_checked
    # eval mode: running statistics, no dropout, no graph
    m.eval()
    with torch.no_grad():
        m.fused_encoder = True
        a1, c1, r1 = m.forward_base(ei, et, [0, n], [0, nr])
        m.fused_encoder = False                                  # (the layers are still op by op from the last _run)
        a2, c2, r2 = m.forward_base(ei, et, [0, n], [0, nr])
    assert_close(a1, a2, 2e-5, 1e-6, "eval align_out")
    assert_close(c1[1], c2[1], 2e-5, 1e-6, "eval c1")
    assert_close(r1[1], r2[1], 2e-5, 1e-6, "eval rel_c1")


def test_fused_encoder_keeps_its_name_block_buffer_across_steps():
    """cat(comp0, info) lives with the model (its constant right block is written once): step after step the same buffer;
    a second forward before the first one's backward gets its own; a backward after the buffer was re-used fails loudly;
    the results are those of a model without the cache."""
    n, nr, d, di = 500, 23, 32, 20
    rng = np.random.default_rng(9)
    ei, et = random_graph(rng, n, nr, 1900, hub=100)
    ei, et = torch.from_numpy(ei).to(DEV), torch.from_numpy(et).to(DEV)
    m = _model(d, n, nr, di, False, 4)
    m.train()
    m.fused_encoder = True
    fwd = lambda: m.forward_base(ei, et, [0, n], [0, nr])
    state = {k: v.clone() for k, v in m.state_dict().items()}
    first = None
    for step in range(3):
        m.load_state_dict(state)                                 # same BN statistics every time: identical outputs
        m.zero_grad(set_to_none=True)
        a, comp, _ = fwd()
        (a.sum() + comp[1].sum()).backward()
        (slot,) = m._encoder_cache["cat0"].values()              # one slot per name-embedding tensor; this model has one
        assert not slot.busy and slot.lease == step + 1
        got = (a.detach().clone(), m.uni_linear1_1.grad.clone())
        if first is None:
            first, buf = got, slot.buf
        else:
            assert slot.buf is buf and torch.equal(got[0], first[0]) and torch.equal(got[1], first[1])
    m.load_state_dict(state)
    a1, comp1, _ = fwd()                                         # holds the model's buffer ...
    a2, _, _ = fwd()                                             # ... so this one allocates its own
    assert slot.buf is buf and slot.busy and len(m._encoder_cache["cat0"]) == 1
    a2.sum().backward()
    assert slot.busy                                             # (not the owner: the first forward still holds it)
    (a1.sum() + comp1[1].sum()).backward(retain_graph=True)      # the owner's backward releases the buffer
    assert not slot.busy
    a3, _, _ = fwd()                                             # re-uses (and rewrites) it
    with pytest.raises(RuntimeError, match="re-used by a later forward"):
        a1.sum().backward()
    del a3


def test_name_block_cache_with_host_name_embeddings_and_equal_size_kgs():
    """ent_info_att stays on the HOST, as in the reference (src/jmac_model.py:133) and in JMAC's constructor here, and the
    model holds two KGs of EQUAL size: every forward_name call used to upload a temporary slice whose address the allocator
    could hand to the other KG's rows -- the cat(comp0, info) cache (keyed on that address) then served the wrong name block.
    The device rows are now persistent per id range and the cache is keyed on them: alternating the two KGs, cached and
    uncached, gives identical outputs, and each KG keeps its own buffer."""
    from jmac_amd.model import JMAC
    n, nr, d, di = 300, 11, 32, 20
    rng = np.random.default_rng(21)
    torch.manual_seed(21)
    args = types.SimpleNamespace(dim=d, dropout=0.0, leaky_relu_w=0.05, comp_op="sub", num_gcn_layer=2, num_negative=4,
                                 margin_align=1.0, margin_completion=5.0, batch_size=8, no_name_info=False, device="cuda")
    m = JMAC(args, rng.standard_normal((2 * n, di)).astype(np.float32), 2 * nr, 2 * n).to(DEV)
    assert not m.ent_info_att.is_cuda                            # the plain attribute did not move with .to()
    graphs = []
    for k in range(2):
        ei, et = random_graph(rng, n, nr, 1200)
        graphs.append((torch.from_numpy(ei).to(DEV), torch.from_numpy(et).to(DEV), [k * n, (k + 1) * n], [k * nr, (k + 1) * nr]))
    m.eval()
    with torch.no_grad():
        ref = []
        for g in graphs:                                         # reference results: the op-by-op path (no cache involved)
            m.fused_encoder = False
            for lay in (m.conv1_alignment, m.conv2_alignment, m.conv1_completion):
                lay.fused = False
            ref.append(m.forward_base(*g)[0].clone())
        m.fused_encoder = True
        for lay in (m.conv1_alignment, m.conv2_alignment, m.conv1_completion):
            lay.fused = True
        for rep in range(3):
            for k in (0, 1, 1, 0):
                junk = torch.randn(n, di, device=DEV)            # churn the allocator between calls
                out = m.forward_base(*graphs[k])[0]
                assert_close(out, ref[k], 2e-5, 1e-6, "kg%d rep %d" % (k, rep))
                del junk
    slots = m._encoder_cache["cat0"]
    assert len(slots) == 2 and len({s_.buf.data_ptr() for s_ in slots.values()}) == 2
    for s_ in slots.values():
        assert s_.info.is_cuda and s_.info.shape == (n, di)


def test_fused_encoder_inference_form_with_bf16_tables():
    """BASELINE config 3's table form through the fused node (no_grad only): bf16 [P|Q|Z] / [Rq|Rz] tables, fp32 arithmetic --
    equal to the op-by-op bf16 path up to the rounding of the relation table (fp32 product rounded once here, a bf16 GEMM
    there), and close to the fp32 tables; with grad mode on the node declines (there is no backward for bf16 tables)."""
    from jmac_amd import encoder
    n, nr, d, di = 700, 37, 32, 20
    rng = np.random.default_rng(5)
    ei, et = random_graph(rng, n, nr, 2600, hub=300)
    ei, et = torch.from_numpy(ei).to(DEV), torch.from_numpy(et).to(DEV)
    m = _model(d, n, nr, di, False, 11)
    m.eval()
    with torch.no_grad():
        m.fused_encoder = True
        a32, c32, _ = m.forward_base(ei, et, [0, n], [0, nr])
        m.set_table_dtype(torch.bfloat16)
        assert encoder.supported(m, di)
        a_f, c_f, r_f = m.forward_base(ei, et, [0, n], [0, nr])
        m.fused_encoder = False
        for lay in (m.conv1_alignment, m.conv2_alignment, m.conv1_completion):
            lay.fused = False
        a_o, c_o, r_o = m.forward_base(ei, et, [0, n], [0, nr])
    with torch.enable_grad():
        assert not encoder.supported(m, di)
    scale = float(a_o.abs().max())
    assert float((a_f - a_o).abs().max()) <= 2e-2 * scale and float((c_f[1] - c_o[1]).abs().max()) <= 2e-2 * float(c_o[1].abs().max())
    assert float((a_f - a32).abs().max()) <= 5e-2 * scale
    assert_close(r_f[1], r_o[1], 2e-5, 1e-6, "rel_c1 (fp32 either way)")


def test_fused_encoder_on_a_slice_of_the_tables():
    """ent_bases / rel_bases select one KG of a multi-KG model (src/jmac_model.py:173-176): the node sees row slices of the
    parameters and autograd scatters its gradients back."""
    n_all, nr_all, d, di = 500, 40, 16, 8
    rng = np.random.default_rng(9)
    m = _model(d, n_all, nr_all, di, False, 4)
    e0, e1, r0, r1 = 120, 420, 20, 40
    n, nr = e1 - e0, r1 - r0
    ei, et = random_graph(rng, n, nr, 900)
    ei, et = torch.from_numpy(ei).to(DEV), torch.from_numpy(et).to(DEV)
    outs, grads = [], []
    for fused in (True, False):
        m.fused_encoder = fused
        for lay in (m.conv1_alignment, m.conv2_alignment, m.conv1_completion):
            lay.fused = fused
        m.zero_grad(set_to_none=True)
        st = {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
        a, c, r = m.forward_base(ei, et, [e0, e1], [r0, r1])
        (a.sum() + (c[1] ** 2).sum() + r[1].sum()).backward()
        outs.append((a.detach(), c[1].detach(), r[1].detach()))
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        m.load_state_dict(st, strict=False)
    for x, y in zip(*outs):
        assert_close(x, y, 2e-5, 1e-6)
    assert set(grads[0]) == set(grads[1])
    scale = max(float(g.abs().max()) for g in grads[1].values())
    for k in grads[1]:
        assert_close(grads[0][k], grads[1][k], 1e-4, (1e-4 if k.endswith("loop_rel") else 1e-6) * scale, k)
    ge = grads[0]["ent_init_att_completion"]
    assert float(ge[:e0].abs().max()) == 0.0 and float(ge[e1:].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(203, 76, 50), (451, 172, 100), (2100, 172, 100)],
                         ids=["203x76x50", "451x172x100", "2100x172x100-tiles64"])
@pytest.mark.parametrize("seed", [0, 1])
def test_grouped_gemm_forms_and_epilogues(seed, shape):
    from jmac_amd.encoder import (ACT_LEAKY, ACT_RELU, DACT_LEAKY, DACT_RELU, gemm_task, grouped_gemm)
    gen = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *s: torch.randn(s, device=DEV, generator=gen)
    M, K, N = shape
    A, A2, B, Bt, At = r(M - 1, K), r(1, K), r(K, N), r(N, K), r(K, M)
    big = r(M, 3 * N)                                             # strided operands / outputs
    src = r(M, N)
    C0 = r(M, N)
    outs = {k: torch.empty(M, N, device=DEV) for k in ("nn", "nt", "tn", "leaky", "relu", "dleaky", "drelu")}
    acc = C0.clone()
    c_hi, c_lo = torch.empty(M - 3, N, device=DEV), torch.empty(3, N, device=DEV)
    acc_hi, acc_lo = C0[:M - 3].clone(), torch.full((3, N), float("nan"), device=DEV)
    strided_out = torch.zeros(M, 3 * N, device=DEV)
    Afull = torch.cat((A, A2))
    tasks = [gemm_task(A, B, outs["nn"], A2=A2),
             gemm_task(Afull, Bt, outs["nt"], tb=True),
             gemm_task(At, B, outs["tn"], ta=True),
             gemm_task(Afull, B, outs["leaky"], act=ACT_LEAKY, slope=0.05),
             gemm_task(Afull, B, outs["relu"], act=ACT_RELU),
             gemm_task(Afull, B, outs["dleaky"], act=DACT_LEAKY, act_src=src, slope=0.05),
             gemm_task(Afull, B, outs["drelu"], act=DACT_RELU, act_src=src),
             gemm_task(Afull, B, acc, accumulate=True),
             gemm_task(Afull, B, c_hi, C2=c_lo),
             gemm_task(Afull, B, acc_hi, C2=acc_lo, accumulate=True),       # accumulate applies to C; C2 rows are stored
             gemm_task(big[:, N:2 * N], r(N, N), strided_out[:, 2 * N:]),
             # transposed A whose K rows continue in a second buffer (the adjoint of cat(rel_emb, loop_rel))
             gemm_task(A, r(M, N), torch.empty(K, N, device=DEV), ta=True, A2=A2)]
    grouped_gemm(tasks)
    ref = Afull.double() @ B.double()
    tol = dict(rtol=1e-5, atol=1e-5)
    assert_close(outs["nn"], ref, **tol)
    assert_close(outs["nt"], Afull.double() @ Bt.double().t(), **tol)
    assert_close(outs["tn"], At.double().t() @ B.double(), **tol)
    assert_close(outs["leaky"], torch.nn.functional.leaky_relu(ref, 0.05), **tol)
    assert_close(outs["relu"], torch.relu(ref), **tol)
    assert_close(outs["dleaky"], torch.where(src > 0, ref, ref * 0.05), **tol)
    assert_close(outs["drelu"], torch.where(src > 0, ref, torch.zeros_like(ref)), **tol)
    assert_close(acc, C0.double() + ref, **tol)
    assert_close(torch.cat((c_hi, c_lo)), ref, **tol)
    assert_close(acc_hi, C0[:M - 3].double() + ref[:M - 3], **tol)
    assert_close(acc_lo, ref[M - 3:], **tol)
    t9, t10 = tasks[10]._keep, tasks[11]._keep
    assert_close(strided_out[:, 2 * N:], t9[0].double() @ t9[1].double(), **tol)
    assert float(strided_out[:, :2 * N].abs().max()) == 0.0       # nothing outside the output slice was touched
    assert_close(t10[2], Afull.double().t() @ t10[1].double(), **tol)
    # bitwise reproducible
    again = torch.empty(M, N, device=DEV)
    grouped_gemm([gemm_task(A, B, again, A2=A2)])
    assert torch.equal(again, outs["nn"])


def test_grouped_gemm_transposed_a_in_two_buffers_any_split():
    """C = cat(A, A2)^T B with the split anywhere in K (the weight-gradient form reads A's rows straight from memory: rows
    past the split come from the second buffer through its general path), odd K, ragged tiles."""
    from jmac_amd.encoder import gemm_task, grouped_gemm
    gen = torch.Generator(device=DEV).manual_seed(3)
    K, M, N = 401, 70, 45
    A, B = torch.randn(K, M, device=DEV, generator=gen), torch.randn(K, N, device=DEV, generator=gen)
    ref = A.double().t() @ B.double()
    for split in (1, 7, 200, 399, 400):
        out = torch.full((M, N), float("nan"), device=DEV)
        grouped_gemm([gemm_task(A[:split].contiguous(), B, out, ta=True, A2=A[split:].contiguous())])
        assert_close(out, ref, 1e-5, 1e-5, "split %d" % split)


def test_grouped_gemm_many_tasks_and_odd_shapes():
    from jmac_amd.encoder import gemm_task, grouped_gemm
    gen = torch.Generator(device=DEV).manual_seed(7)
    tasks, refs = [], []
    for i in range(30):                                           # more than one launch's worth of tasks
        M, K, N = 1 + 37 * i % 211, 1 + 13 * i % 97, 1 + 29 * i % 131
        A, B = torch.randn(M, K, device=DEV, generator=gen), torch.randn(K, N, device=DEV, generator=gen)
        out = torch.empty(M, N, device=DEV)
        tasks.append(gemm_task(A, B, out))
        refs.append((out, A.double() @ B.double()))
    grouped_gemm(tasks)
    for out, ref in refs:
        assert_close(out, ref, 1e-5, 1e-5)


@pytest.mark.parametrize("N,d", [(333, 40), (4099, 300), (2049, 512)], ids=["row-per-wave", "four-rows-per-wave-d300", "four-rows-per-wave-d512"])
@pytest.mark.parametrize("with_mask", [False, True])
def test_normalize_dropout_kernels(with_mask, N, d):
    """(N >= 2048: the forms that keep four rows per wave in registers -- ragged last wave, d = 300 = 75 float4 on 64 + 11 lanes,
    d = 512 = the widest row they take.)"""
    from jmac_amd._lib import check, lib, ptr, stream
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(N, d, device=DEV, generator=gen)
    x[5] = 0.0                                                    # clamped norm
    mask = (torch.rand(N, d, device=DEV, generator=gen) > 0.4).float() if with_mask else None
    scale = 1.0 / 0.6 if with_mask else 1.0
    buf = torch.zeros(N, 3 * d, device=DEV)
    y = buf[:, d:2 * d]
    inv = torch.empty(N, device=DEV)
    check(lib().jmac_row_normalize_drop_fwd_f32(ptr(x), d, N, d, 1e-12, ptr(mask), d, scale, ptr(y), 3 * d, ptr(inv), stream()))
    xr = x.double().requires_grad_(True)
    ref = torch.nn.functional.normalize(xr, 2, -1)
    if with_mask:
        ref = ref * mask.double() * scale
    assert_close(y, ref.detach(), 1e-6, 1e-7)
    assert float(buf[:, :d].abs().max()) == 0.0 and float(buf[:, 2 * d:].abs().max()) == 0.0
    gbuf = torch.randn(N, 2 * d, device=DEV, generator=gen)
    g = gbuf[:, d:]
    ref.backward(g.double())
    gx = torch.full((N, d), 7.0, device=DEV)
    base = torch.randn(N, d, device=DEV, generator=gen)
    gacc = base.clone()
    for out, accumulate in ((gx, 0), (gacc, 1)):
        check(lib().jmac_row_normalize_drop_bwd_f32(ptr(x), d, ptr(inv), ptr(mask), d, scale, ptr(g), 2 * d, N, d, 1e-12, ptr(out),
                                                    d, accumulate, stream()))
    want = xr.grad.clone()
    want[5] = (g[5].double() * (mask[5].double() * scale if with_mask else 1.0)) * 1e12   # ||x|| <= eps: y = x / eps
    assert_close(gx[torch.arange(N) != 5], want[torch.arange(N) != 5], 1e-5, 1e-6)
    assert_close(gx[5], want[5], 1e-5)
    assert_close(gacc[torch.arange(N) != 5], (base.double() + want)[torch.arange(N) != 5], 1e-5, 1e-6)


@pytest.mark.parametrize("p_drop", [0.4, 0.05])
def test_normalize_dropout_seeded_draws(p_drop):
    """The in-kernel Bernoulli draws: (1) the forward equals the mask form run with the mask read back from its own zeros,
    bit for bit; (2) the backward regenerates exactly those draws; (3) the keep rate is 1 - p within sampling error, draws
    of different seeds / rows / columns are uncorrelated; (4) the same seed gives the same draws (a captured step replays
    the kernel, the seed tensor's CONTENT is what changes)."""
    from jmac_amd._lib import check, lib, ptr, stream
    gen = torch.Generator(device=DEV).manual_seed(11)
    N, d = 2050, 300                                              # (>= 2048 rows: the four-rows-per-wave forms, ragged last wave)
    x = torch.randn(N, d, device=DEV, generator=gen) + 3.0        # no exact zeros among the normalised values
    seed = torch.tensor([0x1234_5678_9ABC_DEF0 >> 1, 77], dtype=torch.int64, device=DEV)
    L = lib()

    def fwd(sd):
        y, inv = torch.empty(N, d, device=DEV), torch.empty(N, device=DEV)
        check(L.jmac_row_normalize_dropseed_fwd_f32(ptr(x), d, N, d, 1e-12, ptr(sd), p_drop, ptr(y), d, ptr(inv), stream()))
        return y, inv
    y, inv = fwd(seed[0:1])
    mask = (y != 0).float()
    keep = float(mask.mean())
    assert abs(keep - (1 - p_drop)) < 4 * (p_drop * (1 - p_drop) / (N * d)) ** 0.5 + 1e-4, keep
    scale = 1.0 / (1.0 - p_drop)
    y2, inv2 = torch.empty(N, d, device=DEV), torch.empty(N, device=DEV)
    check(L.jmac_row_normalize_drop_fwd_f32(ptr(x), d, N, d, 1e-12, ptr(mask), d, scale, ptr(y2), d, ptr(inv2), stream()))
    assert torch.equal(y, y2) and torch.equal(inv, inv2)
    assert torch.equal(fwd(seed[0:1])[0], y)                      # same seed, same draws
    other = (fwd(seed[1:2])[0] != 0).float()
    assert abs(float((mask * other).mean()) - (1 - p_drop) ** 2) < 5e-3          # another seed: independent draws
    assert abs(float((mask[1:] * mask[:-1]).mean()) - (1 - p_drop) ** 2) < 5e-3  # neighbouring rows
    assert abs(float((mask[:, 1:] * mask[:, :-1]).mean()) - (1 - p_drop) ** 2) < 5e-3
    g = torch.randn(N, d, device=DEV, generator=gen)
    base = torch.randn(N, d, device=DEV, generator=gen)
    for acc in (0, 1):
        ga, gb = base.clone(), base.clone()
        check(L.jmac_row_normalize_dropseed_bwd_f32(ptr(x), d, ptr(inv), ptr(seed[0:1]), p_drop, ptr(g), d, N, d, 1e-12, ptr(ga), d,
                                                    acc, stream()))
        check(L.jmac_row_normalize_drop_bwd_f32(ptr(x), d, ptr(inv), ptr(mask), d, scale, ptr(g), d, N, d, 1e-12, ptr(gb), d, acc,
                                                stream()))
        assert_close(ga, gb.double(), 1e-6, 1e-7)               # same draws (the two kernels contract their FMAs differently)
    bad = torch.empty(N, d, device=DEV)
    assert L.jmac_row_normalize_dropseed_fwd_f32(ptr(x), d, N, d, 1e-12, ptr(seed), 1.0, ptr(bad), d, ptr(inv), stream()) != 0


def test_bn_tanh_two_destinations_two_gradients():
    from jmac_amd import ops
    from jmac_amd.encoder import _bn_bwd, _bn_fwd
    gen = torch.Generator(device=DEV).manual_seed(2)
    N, d = 517, 24
    x = torch.randn(N, d, device=DEV, generator=gen) * 0.3 + 0.1
    bn = torch.nn.BatchNorm1d(d).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.2, 0.2)
    b1, b2 = torch.zeros(N, 2 * d, device=DEV), torch.zeros(N, 3 * d, device=DEV)
    mean, invstd, use_batch = _bn_fwd(x, bn, True, b1[:, d:], b2[:, :d])
    xr = x.clone().requires_grad_(True)
    bn2 = torch.nn.BatchNorm1d(d).to(DEV)
    bn2.load_state_dict({k: v for k, v in bn.state_dict().items()}, strict=True)
    bn2.running_mean.zero_(); bn2.running_var.fill_(1.0); bn2.num_batches_tracked.zero_()
    ref = torch.tanh(bn2(xr))
    assert_close(b1[:, d:], ref.detach(), 1e-5, 1e-6)
    assert torch.equal(b1[:, d:], b2[:, :d])
    assert_close(bn.running_mean, bn2.running_mean, 1e-5, 1e-7)
    assert_close(bn.running_var, bn2.running_var, 1e-5, 1e-7)
    assert int(bn.num_batches_tracked) == 1
    g1b, g2 = torch.randn(N, 2 * d, device=DEV, generator=gen), torch.randn(N, d, device=DEV, generator=gen)
    g1 = g1b[:, :d]
    gx, gbw = _bn_bwd(x, b1[:, d:], g1, g2, bn.weight, mean, invstd, use_batch)
    ref.backward(g1 + g2)
    assert_close(gx, xr.grad, 1e-4, 1e-6)
    assert_close(gbw[:d], bn2.bias.grad, 1e-4, 1e-6)
    assert_close(gbw[d:], bn2.weight.grad, 1e-4, 1e-6)


@pytest.mark.parametrize("relu", [False, True], ids=["leaky", "dbpv1-relu"])
def test_layer_node_equals_op_by_op_layer(relu):
    """RelationAwareLayer.forward as one node (encoder._LayerNode) against the op-by-op layer: output, running statistics and
    every gradient, both relation activations (src/jmac_model.py:41 LeakyReLU, JMAC_DBPv1/models/jmac_model.py:51 ReLU)."""
    from jmac_amd.layer import RelationAwareLayer, RelationalAwareLayer
    from util import make_args
    n, nr, d = 900, 23, 40
    rng = np.random.default_rng(3)
    ei, et = random_graph(rng, n, nr, 3000, hub=280)
    ei, et = torch.from_numpy(ei).to(DEV), torch.from_numpy(et).to(DEV)
    torch.manual_seed(5)
    lay = (RelationalAwareLayer(d, d, nr, rel_dim=d, act=torch.tanh, args=make_args()) if relu
           else RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args())).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(8)
    X0 = torch.randn(n, d, device=DEV, generator=gen) * 0.4
    R0 = torch.randn(nr, d, device=DEV, generator=gen) * 0.4
    G = torch.randn(n, d, device=DEV, generator=gen)
    res = []
    for fused in (True, False):
        lay.fused = fused
        lay.zero_grad(set_to_none=True)
        lay.bn.running_mean.zero_(); lay.bn.running_var.fill_(1.0); lay.bn.num_batches_tracked.zero_()
        X, R = X0.clone().requires_grad_(True), R0.clone().requires_grad_(True)
        lay.train()
        out = lay(X, R, ei, et)
        (out * G).sum().backward()
        lay.eval()
        with torch.no_grad():
            out_eval = lay(X, R, ei, et)
        res.append((out.detach(), out_eval, X.grad, R.grad, lay.bn.running_mean.clone(), lay.bn.running_var.clone(),
                    {k: p.grad.clone() for k, p in lay.named_parameters()}))
    a, b = res
    for x, y, what in zip(a[:6], b[:6], ("out", "out_eval", "grad X", "grad R", "running_mean", "running_var")):
        assert_close(x, y, 1e-4, 1e-6, what)
    scale = max(float(g.abs().max()) for g in b[6].values())
    for k in b[6]:
        assert_close(a[6][k], b[6][k], 1e-4, (1e-4 if k == "loop_rel" else 1e-6) * scale, k)


@pytest.mark.parametrize("no_name", [False, True], ids=["forward_name", "forward_no_name"])
def test_relation_side_on_used_rows_equals_all_rows(no_name):
    """encoder._RelCompact: the layers' relation chains run on the relation rows the graph's edges name (a DBP-5L KG: 153-833 of
    961) -- same outputs and the same gradients as with every row (rows no edge names reach no output; their gradient is the
    MLP's alone for rel_init_att_completion and exactly zero for rel_init_att_alignment), for each pattern of used outputs."""
    from jmac_amd import encoder
    n, nr, d, di = 600, 61, 32, 20
    rng = np.random.default_rng(17)
    ei, et = random_graph(rng, n, 9, 2400, hub=150)
    used = np.array([3, 7, 8, 20, 21, 40, 41, 42, 60])                  # 9 of the 61 relation rows carry edges
    et = used[et]
    ei, et = torch.from_numpy(ei).to(DEV), torch.from_numpy(et).to(DEV)
    m = _model(d, n, nr, di, no_name, 23)
    m.train()
    gen = torch.Generator(device=DEV).manual_seed(1)
    G = [torch.randn(n, d, device=DEV, generator=gen), torch.randn(n, d, device=DEV, generator=gen),
         torch.randn(nr, d, device=DEV, generator=gen)]
    state = {k: v.clone() for k, v in m.state_dict().items()}
    res = {}
    for use in ((True, True, True), (False, True, True), (True, False, False)):
        for flag in (True, False):
            encoder.COMPACT_RELATIONS = flag
            try:
                m.load_state_dict(state)
                m.zero_grad(set_to_none=True)
                a, comp, rel = m.forward_base(ei, et, [0, n], [0, nr])
                loss = 0
                if use[0] and not no_name:
                    loss = loss + (a * G[0]).sum()
                if use[1]:
                    loss = loss + (comp[1] * G[1]).sum()
                if use[2]:
                    loss = loss + (rel[1] * G[2]).sum()
                if not torch.is_tensor(loss):
                    continue
                loss.backward()
                res[(use, flag)] = (float(loss.detach()), [x.detach().clone() for x in (a, comp[1], rel[1])],
                                    {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()})
            finally:
                encoder.COMPACT_RELATIONS = True
        if (use, True) not in res:
            continue
        (l1, o1, g1), (l0, o0, g0) = res[(use, True)], res[(use, False)]
        assert abs(l1 - l0) <= 1e-5 * abs(l0)
        for x, y in zip(o1, o0):
            assert_close(x, y, 1e-5, 1e-6, "outputs %s" % (use,))
        gscale = max(float(g.abs().max()) for g in g0.values() if g is not None)
        for k in g0:
            if g0[k] is None:
                assert g1[k] is None or float(g1[k].abs().max()) == 0.0, k
                continue
            assert_close(g1[k], g0[k], 2e-5, 1e-6 * gscale, "grad %s %s" % (k, use))
        unused = np.setdiff1d(np.arange(nr), used)
        if not no_name and g1["rel_init_att_alignment"] is not None:
            assert float(g1["rel_init_att_alignment"][unused].abs().max()) == 0.0       # exactly zero on rows no edge names


@pytest.mark.parametrize("use", [("align", "comp", "rel"), ("comp", "rel")])
def test_active_row_projections_equal_the_full_products(use, monkeypatch):
    """Round 5: inside the encoder node the entities are in CLASS order (destination only | both | source only | neither) and the
    projections run on row ranges -- P for the destinations, Q for the sources, Z for every row (encoder._RowOrder;
    src/jmac_model.py:75-76,85).  Outputs, every gradient and the BatchNorm buffers must equal the full-product form
    (ACTIVE_ROWS = False) to rounding, on a graph with all four classes; and nothing may ever READ a P / Q row that was not
    written: the [N, 3d] tables are poisoned with NaN before the products write their row ranges, so one such read that reaches
    any arithmetic would surface as NaN in an output or a gradient (it found one: pass A of the backward multiplied the clamped
    loads of the self row's Q half by zero instead of selecting them away)."""
    from jmac_amd import encoder
    n, nr, d, di = 6000, 37, 64, 20
    rng = np.random.default_rng(7)
    e = 9000
    dst = rng.choice(n // 2, size=e)                              # half of the entities are never destinations
    src = rng.choice(np.arange(n // 4, n * 3 // 4), size=e)       # a different half are never sources; a quarter neither
    dst[:300] = 11                                                # a hub
    ei = torch.from_numpy(np.stack([dst, src]).astype(np.int64)).to(DEV)
    et = torch.from_numpy(rng.integers(0, nr, e).astype(np.int64)).to(DEV)
    m = _model(d, n, nr, di, False, 13)
    gen = torch.Generator(device=DEV).manual_seed(4)
    G = {k: torch.randn(s, device=DEV, generator=gen) for k, s in
         (("align", (n, d)), ("c1", (n, d)), ("c0", (n, d)), ("r1", (nr, d)), ("r0", (nr, d)))}
    m.train()
    real_empty = encoder._empty

    def poisoned(dev, *shape):                                   # the [P|Q|Z] tables (and their gradients: fully written anyway)
        t_ = real_empty(dev, *shape)
        return t_.fill_(float("nan")) if tuple(shape) == (n, 3 * d) else t_
    monkeypatch.setattr(encoder, "_empty", poisoned)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(encoder, "ACTIVE_ROWS", flag)
        res[flag] = _run(m, True, ei, et, n, nr, use, G)
        if flag:
            from jmac_amd.graph import graph_cache
            g = graph_cache.get(ei, et, n, nr + 1, m.conv1_completion.chunk)
            ro = g._row_orders[None]
            assert 0 < ro.s0 < ro.nD < ro.s1 < n and ro.fraction < 0.6          # all four classes; the path was taken
    (out_a, g_a, bn_a), (out_f, g_f, bn_f) = res[True], res[False]
    for a, b, what in zip(out_a, out_f, ("align_out", "c1", "rel_c1")):
        assert bool(torch.isfinite(a).all()), what
        assert_close(a, b, 2e-5, 1e-6, what)
    for k in bn_f:
        assert_close(bn_a[k], bn_f[k], 1e-5, 1e-7, k)
    scale = max(float(g.abs().max()) for g in g_f.values() if g is not None)
    for k, ref in g_f.items():
        got = g_a[k]
        if ref is None:
            assert got is None or float(got.abs().max()) == 0.0, k
            continue
        assert got is not None and bool(torch.isfinite(got).all()), k
        atol = 1e-4 * scale if k.endswith("loop_rel") else 1e-6 * scale
        assert_close(got, ref, 1e-4, atol, "grad " + k)


def test_paired_first_layer_launches_equal_two_launches_bitwise():
    """Round 5: conv1_alignment and conv1_completion of forward_name (src/jmac_model.py:183,190) are independent and share the
    graph; their forward aggregations run as ONE launch (jmac_rel_attn_aggregate_fwd_jobs_f32).  Same body, same per-row order:
    outputs and gradients are those of the two-launch form, bit for bit."""
    from jmac_amd import encoder
    n, nr, d, di = 5000, 37, 300, 20
    rng = np.random.default_rng(9)
    ei, et = random_graph(rng, n, nr, 12000, hub=300)
    ei, et = torch.from_numpy(ei).to(DEV), torch.from_numpy(et).to(DEV)
    m = _model(d, n, nr, di, False, 17)
    gen = torch.Generator(device=DEV).manual_seed(5)
    G = {k: torch.randn(s, device=DEV, generator=gen) for k, s in
         (("align", (n, d)), ("c1", (n, d)), ("c0", (n, d)), ("r1", (nr, d)), ("r0", (nr, d)))}
    m.train()
    res = {}
    for flag in (True, False):
        encoder.PAIR_LAUNCHES = flag
        try:
            res[flag] = _run(m, True, ei, et, n, nr, ("align", "comp", "rel"), G)
        finally:
            encoder.PAIR_LAUNCHES = True
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    for k, ref in res[False][1].items():
        got = res[True][1][k]
        assert (ref is None and got is None) or torch.equal(got, ref), k
