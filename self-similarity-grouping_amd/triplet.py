"""Pairwise block of the fine-tune phase's TripletLoss on the GPU -- reid/loss/triplet.py:28-31:

    dist = torch.pow(inputs, 2).sum(dim=1, keepdim=True).expand(n, n)
    dist = dist + dist.t()
    dist.addmm_(1, -2, inputs, inputs.t())
    dist = dist.clamp(min=1e-12).sqrt()

Forward: the fp32-MFMA Gram kernel with its distance epilogue (`ssg_pairwise_sqdist_f32`, shared with the evaluator's
`pairwise_distance`) + an in-place clamp/sqrt.  Backward (round 4): `pairwise_dist` is a `torch.autograd.Function`, so the block can
stand where the reference's four lines stand inside `TripletLoss.forward` -- the loss back-propagates through `dist` into the
features: grad_x = diag(rowsum(S)) x - S x with S = W + W^T, W = grad_dist / dist where the clamp passes the gradient; S x runs on
the same fp32-MFMA GEMM (`ssg_conv2d_nhwc_f32` as a 1 x 1 convolution), the two elementwise halves are HIP kernels
(`ssg_triplet_grad_weights`, `ssg_triplet_grad_combine`).  The loss's mining loops stay in the reference's Python."""
import torch

from . import _lib
from ._lib import check, ptr, stream
from .evaluators import _sqdist


class _PairwiseDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, clamp_min):
        L = _lib.lib()
        xc = x.detach().to(torch.device("cuda", torch.cuda.current_device()), torch.float32).contiguous()
        sq = _sqdist(xc, xc).contiguous()
        dist = sq.clone()
        check(L.ssg_clamp_sqrt_f32(ptr(dist), dist.numel(), float(clamp_min), stream()), "ssg_clamp_sqrt_f32")
        ctx.save_for_backward(xc, sq, dist)
        ctx.clamp_min, ctx.in_device, ctx.in_dtype = float(clamp_min), x.device, x.dtype
        return dist

    @staticmethod
    def backward(ctx, grad_dist):
        L = _lib.lib()
        x, sq, dist = ctx.saved_tensors
        n, d = x.shape
        dev, st = x.device, stream()
        g = grad_dist.to(dev, torch.float32).contiguous()
        ld = (n + 31) // 32 * 32                       # K granule of the GEMM
        dp = (d + 63) // 64 * 64                       # its output-channel granule
        S = torch.empty((n, ld), dtype=torch.float32, device=dev)
        rowsum = torch.empty(n, dtype=torch.float32, device=dev)
        check(L.ssg_triplet_grad_weights(ptr(g), ptr(sq), ptr(dist), n, ld, ctx.clamp_min, ptr(S), ptr(rowsum), st), "ssg_triplet_grad_weights")
        xt = torch.zeros((dp, ld), dtype=torch.float32, device=dev)
        xt[:d, :n] = x.t()                             # the GEMM's "weights": x^T, one row per feature channel
        zeros = torch.zeros(dp, dtype=torch.float32, device=dev)
        Sx = torch.empty((n, dp), dtype=torch.float32, device=dev)
        check(L.ssg_conv2d_nhwc_f32(ptr(S), ptr(xt), ptr(zeros), None, ptr(Sx), n, 1, 1, ld, dp, 1, 1, 1, 0, 0, st), "ssg_conv2d_nhwc_f32 (S x)")
        gx = torch.empty((n, d), dtype=torch.float32, device=dev)
        check(L.ssg_triplet_grad_combine(ptr(x), ptr(rowsum), ptr(Sx), n, d, dp, ptr(gx), st), "ssg_triplet_grad_combine")
        return gx.to(device=ctx.in_device, dtype=ctx.in_dtype), None


def pairwise_dist(inputs, clamp_min=1e-12):
    """inputs [n, d] float32 (any device; may require grad) -> [n, n] float32 CUDA, float32 accuracy (GEMM accumulation order differs
    from torch's).  Differentiable: the gradient flows back to `inputs` like through the reference's four lines."""
    x = torch.as_tensor(inputs)
    if x.dim() != 2:
        raise ValueError("inputs must be [n, d]")
    return _PairwiseDist.apply(x, clamp_min)
