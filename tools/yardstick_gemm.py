#!/usr/bin/env python3
"""Yardstick only (not part of the product): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, fp16 in, fp32 accumulate) reaches on
this box for the GEMM shapes behind the embedding's plain launches and the source term's bound pass, on random operands and on operands
with half of the A elements zero (post-ReLU statistics) -- next to the EXECUTED fp16 rate of this repo's kernels on the same shapes
(three MFMA products per fp32 multiply: executed = 3 x the fp32-equivalent figure of tools/layer_table.py)."""
import sys, os
import torch
dev = torch.device("cuda", 0)
shapes = [("layer3 conv1 (1x1 1024->256), M=128000", 128000, 256, 1024, 0.196 * 1e-3, 3),
          ("layer3 3x3 256->256 as a GEMM, K=2304", 128000, 256, 2304, 0.350 * 1e-3, 3),
          ("layer3 conv3 (1x1 256->1024)", 128000, 1024, 256, 0.317 * 1e-3, 3),
          ("layer4 conv1 (1x1 2048->512), M=32000", 32000, 512, 2048, 0.172 * 1e-3, 3),
          ("source-term bound pass 16000 x 12936 x 2048", 16000, 12936, 2048, 0.84 * 1e-3, 1)]
for name, M, N, K, ours_s, prods in shapes:
    res = []
    for zero_frac in (0.0, 0.5):
        a = torch.randn(M, K, device=dev, dtype=torch.float16)
        if zero_frac:
            a = a * (torch.rand(M, K, device=dev) >= zero_frac).half()
        b = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.05
        for _ in range(3):
            c = a @ b.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            c = a @ b.t()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append(2.0 * M * N * K / ms / 1e9)
    ours = prods * 2.0 * M * N * K / ours_s / 1e12
    print("%-48s vendor fp16 GEMM %6.0f TFLOP/s (random) %6.0f (A half zero) | this repo executes %6.0f TFLOP/s fp16 on the shape (%d product%s%s)" % (
        name, res[0], res[1], ours, prods, "s" if prods > 1 else "", ", + residual / epilogue traffic" if "conv3" in name else ""), flush=True)
