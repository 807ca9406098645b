#!/usr/bin/env python3
"""Encode / k-means assignment kernel (annlite/core/codec/pq.py:158-177 -> encode_kernel, codec.hip): time per launch and the
fraction of the chip's fp32 vector rate it reaches.  One JSON line per shape (profiles/r03/encode.jsonl).

Work per (row, sub-space, codeword): dsub x (one subtract + one fused multiply-add) -- the reference's arithmetic restated as
the fmaf chain the oracle pins (first minimum wins).  SURVEY.md section 8d counts it as the GEMM it could be, 2 N D Ks flops;
both conventions are reported: `gemm_equivalent` against the 157.3 TFLOP/s vector peak, and `valu_issue` = executed vector
instructions (2 per element pair + the compare / select per codeword) against the issue rate of one wave64 instruction per
2 clocks per SIMD -- the roof of THIS formulation.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annlite_amd import ops  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
g = torch.Generator(device=dev)
g.manual_seed(0)
for N, D, M in ((1_000_000, 128, 16), (250_000, 768, 64), (1_000_000, 128, 8)):
    Ks, dsub = 256, D // M
    x = torch.randn((N, D), generator=g, device=dev)
    cb = torch.randn((M, Ks, dsub), generator=g, device=dev)
    ops.pq_encode(x, cb)
    torch.cuda.synchronize()
    ms = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.pq_encode(x, cb)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    t = float(np.median(ms)) * 1e-3
    gemm = 2.0 * N * D * Ks
    instr = N * M * Ks * (2.0 * dsub + 2.0) / 64.0            # wave64 vector instructions
    issue_peak = 256 * 4 * 2.4e9 / 2.0                        # wave64 instructions per second (one per 2 clocks per SIMD)
    print(json.dumps({'kernel': 'encode_kernel', 'rows': N, 'dim': D, 'm': M, 'ks': Ks, 'ms': t * 1e3, 'rows_per_s': N / t,
                      'gemm_equivalent': {'tflops': gemm / t / 1e12, 'peak_tflops': 157.3, 'frac': gemm / t / 1e12 / 157.3},
                      'valu_issue': {'instr_per_s': instr / t, 'peak': issue_peak, 'frac': instr / t / issue_peak},
                      'hbm': {'GB_per_s': (N * D * 4 + N * M) / t / 1e9, 'frac': (N * D * 4 + N * M) / t / 1e9 / 8000.0}}))
