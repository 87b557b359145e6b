"""Pairwise block of the fine-tune phase's TripletLoss on the GPU -- reid/loss/triplet.py:28-31:

    dist = torch.pow(inputs, 2).sum(dim=1, keepdim=True).expand(n, n)
    dist = dist + dist.t()
    dist.addmm_(1, -2, inputs, inputs.t())
    dist = dist.clamp(min=1e-12).sqrt()

as the fp32-MFMA Gram kernel with its distance epilogue (`ssg_pairwise_sqdist_f32`, shared with the evaluator's
`pairwise_distance`) + an in-place clamp/sqrt.  Forward only: the loss's mining loops and the backward pass belong to the
training phase, outside the grouping hot path."""
import torch

from . import _lib
from ._lib import check, ptr, stream
from .evaluators import _sqdist


def pairwise_dist(inputs, clamp_min=1e-12):
    """inputs [n, d] float32 (any device) -> [n, n] float32 CUDA, float32 accuracy (GEMM accumulation order differs from torch's)."""
    L = _lib.lib()
    x = torch.as_tensor(inputs)
    if x.dim() != 2:
        raise ValueError("inputs must be [n, d]")
    d2 = _sqdist(x, x).contiguous()
    check(L.ssg_clamp_sqrt_f32(ptr(d2), d2.numel(), float(clamp_min), stream()), "ssg_clamp_sqrt_f32")
    return d2
