#!/usr/bin/env python3
"""Where a K-step of the tiled GEMM spends its cycles: a diagnostic build with s_memtime stamps
around the loop's segments (wait for the stage's DMA, barrier, issue of the next stage's DMA,
fragment reads, MFMAs); shares per segment, median over waves.  Not a timing run."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kvmini.ops import native

C = native()
dev = torch.device("cuda:0")
out = []
for (M, N, K) in ((2048, 28672, 4096), (2048, 4096, 4096)):
    W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    wp = C.pack_weight(W)
    x = torch.randn(M, K, device=dev).bfloat16()
    y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for variant, waves in ((90, 8), (92, 8), (95, 4)):
        nst = K // (32 * C.wide_gemm_stage_ksteps(256, 256, variant))
        nwg = (M // 256) * (N // 256)
        dbg = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
        for _ in range(3):
            C.wide_gemm(wp, N, K, x, y, 0, 256, 256, variant, 1, nst, None, 0.0, None, False, dbg)
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(nwg, 8, 8)[:, :waves, :5].astype(np.float64)
        per_step = np.median(d.reshape(-1, 5), axis=0) / nst
        tot = per_step.sum()
        rec = {"M": M, "N": N, "K": K, "variant": variant, "waves": waves, "stages": nst,
               "cycles_per_stage": {k: round(float(v), 1) for k, v in zip(("wait_dma", "barrier", "issue_dma", "frag_reads", "mfma"), per_step)},
               "share": {k: round(float(v / tot), 3) for k, v in zip(("wait_dma", "barrier", "issue_dma", "frag_reads", "mfma"), per_step)}, "total": round(float(tot), 1)}
        out.append(rec)
        print(json.dumps(rec), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/wide_gemm_stamps.json", "w"), indent=1)
