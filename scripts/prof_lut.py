#!/usr/bin/env python3
"""LUT construction only (the command rocprofv3 wraps for the MFMA evidence): inner-product tables on the
matrix cores (cosine / inner-product metrics) and the L2 fmaf-chain tables, at the C2/C3 and C4 shapes."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annlite_amd import ops  # noqa: E402
from annlite_amd._capi import LAYOUT_TILED, LUT_IPDIST, LUT_L2  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument('--iters', type=int, default=20)
a = p.parse_args()
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
for name, B, M, dsub, qi in (('c3_m16_d128_b1024', 1024, 16, 8, 4), ('c4_m64_d768_b256', 256, 64, 12, 2)):
    D, Ks = M * dsub, 256
    cb = torch.randn((M, Ks, dsub), generator=g, device=dev)
    q = torch.randn((B, D), generator=g, device=dev)
    for kind, kname in ((LUT_IPDIST, 'ipdist'), (LUT_L2, 'l2')):
        for _ in range(3):
            ops.lut_build(q, cb, kind, LAYOUT_TILED, qi)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(a.iters):
            ops.lut_build(q, cb, kind, LAYOUT_TILED, qi)
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / a.iters
        flops = 2.0 * B * D * Ks
        print('%s %-6s %8.1f us/build  %7.2f TFLOP/s (2*B*D*Ks = %.0f MFLOP)  %6.1f GB/s table writes' %
              (name, kname, us, flops / us / 1e6, flops / 1e6, B * M * Ks * 4 / us / 1e3))
