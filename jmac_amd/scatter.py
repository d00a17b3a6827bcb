"""torch_scatter-compatible trio on libjmac_hip.so (boundary B2).

``scatter_add`` / ``scatter`` / ``scatter_softmax`` with the signatures the reference calls
(src/jmac_model.py:105; modules/helper/message_passing.py:24,28).  Putting this module on the path as
``torch_scatter`` lets the UNMODIFIED reference layer run on HIP kernels (un-fused; the fast path is
``jmac_amd.layer.RelationAwareLayer``).  Differentiable like the originals.
"""
from __future__ import annotations

import torch

from ._lib import check, lib, ptr, require_device, stream


def _prep(src: torch.Tensor, index: torch.Tensor, dim: int):
    if dim not in (0, -src.dim()):
        raise NotImplementedError("jmac_amd.scatter supports dim=0 only (all the reference uses)")
    require_device(src, index)
    if src.dtype != torch.float32:
        raise TypeError("fp32 only")
    index = index.reshape(-1).to(torch.int64).contiguous()
    E = src.shape[0]
    if index.numel() != E:
        raise ValueError("index must have one entry per row of src")
    width = 1
    for k in src.shape[1:]:
        width *= int(k)
    return src.contiguous().reshape(E, width), index                  # (E = 0: "-1" would be ambiguous)


class _ScatterSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src2d, index, n):
        E, d = src2d.shape
        out = torch.zeros((n, d), dtype=torch.float32, device=src2d.device)
        check(lib().jmac_scatter_sum_f32(ptr(src2d), ptr(index), E, d, n, ptr(out), stream()), "jmac_scatter_sum_f32")
        ctx.save_for_backward(index)
        return out

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        return g.index_select(0, index), None, None


class _ScatterSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src2d, index, n):
        E, d = src2d.shape
        L = lib()
        out = torch.empty_like(src2d)
        ws_bytes = int(L.jmac_scatter_softmax_workspace_bytes(n, d))
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=src2d.device)
        check(L.jmac_scatter_softmax_f32(ptr(src2d), ptr(index), E, d, n, ptr(out), ptr(ws), ws_bytes, stream()),
              "jmac_scatter_softmax_f32")
        ctx.save_for_backward(out, index)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        y, index = ctx.saved_tensors
        gy = (g * y).contiguous()
        s = _ScatterSum.apply(gy, index, ctx.n)
        return y * (g - s.index_select(0, index)), None, None


def _dim_size(index, dim_size):
    if dim_size is not None:
        return int(dim_size)
    return int(index.max().item()) + 1 if index.numel() else 0


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    if out is not None:
        raise NotImplementedError("out= is not used by the reference")
    src2d, idx = _prep(src, index, dim)
    res = _ScatterSum.apply(src2d, idx, _dim_size(idx, dim_size))
    return res.reshape((res.shape[0],) + tuple(src.shape[1:]))


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    if reduce not in ("sum", "add"):
        raise NotImplementedError("the reference only reaches reduce='sum' (message_passing.py:22)")
    return scatter_add(src, index, dim, out, dim_size)


def scatter_softmax(src, index, dim=0, dim_size=None):
    src2d, idx = _prep(src, index, dim)
    res = _ScatterSoftmax.apply(src2d, idx, _dim_size(idx, dim_size))
    return res.reshape(src.shape)
