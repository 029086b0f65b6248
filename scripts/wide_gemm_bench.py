#!/usr/bin/env python3
"""Tiled MFMA GEMM (csrc/wide_gemm.hip) on one MI355X: a correctness sweep over every tile
shape / epilogue against an fp32 matmul, then interleaved timings against the library GEMM
(hipBLASLt through ``F.linear``) on the Llama-3-8B projection shapes, on the same random data,
weights rotated through several copies so the Infinity Cache cannot serve re-reads.

    python scripts/wide_gemm_bench.py --out gpurun_out/wide_gemm.json [--rows 256,512,1024,2048] [--fp8]
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.nn.functional as F

TILES = ((256, 256), (256, 224), (256, 192), (256, 128), (256, 64), (128, 128), (128, 256))
# pipeline variants compiled per tile (csrc/wide_gemm.hip::dispatch_tile); FP8 has no 256 x 64 tile and no 32-wide stages
VARIANTS = {(256, 256): (0, 1, 3), (256, 224): (0, 1, 3), (256, 192): (0, 1, 3), (256, 128): (0, 1, 2, 4, 5), (128, 256): (0, 2, 4), (256, 64): (0, 2), (128, 128): (0, 2, 4, 5)}
VARIANTS_FP8 = {(256, 256): (0,), (256, 224): (0,), (256, 192): (0,), (256, 128): (0, 2, 4, 5), (128, 256): (0, 2, 4), (128, 128): (0, 2, 4, 5)}


def time_fns(fns, rounds=8, inner=4):
    res = {k: [] for k in fns}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                f()
            e1.record()
            e1.synchronize()
            res[k].append(e0.elapsed_time(e1) * 1000.0 / inner)
    return {k: (statistics.median(v), min(v)) for k, v in res.items()}


def check(C, dev, log):
    """Every tile shape x epilogue (bf16 with row scale, slabs with split-K, SwiGLU), bf16 and FP8
    weights, M not a multiple of the tile: max error against fp32."""
    from kvmini.models.llama import quantize_fp8_rows

    torch.manual_seed(0)
    worst = 0.0
    for (bm, bn), variant in [(t, v) for t in TILES for v in VARIANTS[t]]:
        for M, N, K in ((200, 512, 256), (513, 256, 1024)) if bn in (256, 128, 64) else ((200, 2 * bn, 256), (513, bn, 1024)):
            x = torch.randn(M, K, device=dev).bfloat16()
            W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
            ss = (x.float() ** 2).sum(-1, keepdim=True).contiguous()
            rs = torch.rsqrt(ss / K + 1e-5)
            for fp8 in (False, True):
                if fp8 and variant not in VARIANTS_FP8.get((bm, bn), ()):
                    continue
                nst = K // (32 * C.wide_gemm_stage_ksteps(bm, bn, variant))
                if fp8:
                    q, scale = quantize_fp8_rows(W)
                    wp = C.pack_weight_fp8(q.view(torch.uint8).contiguous())
                    wref = q.float() * scale[:, None]
                    ws = scale.contiguous()
                else:
                    wp, wref, ws = C.pack_weight(W), W.float(), None
                ref = x.float() @ wref.T
                # bf16 + row scale
                y = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
                C.wide_gemm(wp, N, K, x, y, 0, bm, bn, variant, 1, nst, ss, 1e-5, ws, True)
                e0 = ((y.float() - ref * rs).abs().max() / (ref * rs).abs().max()).item()
                # slabs, 2 splits (uneven when K/64 is odd-divisible), n fastest
                sps = (nst + 2) // 3  # three splits, the last one shorter or as long
                nsp = -(-nst // sps)
                sl = torch.full((nsp, M, N), float("nan"), device=dev, dtype=torch.float32)
                C.wide_gemm(wp, N, K, x, sl, 1, bm, bn, variant, nsp, sps, None, 0.0, ws, False)
                e1 = ((sl.sum(0) - ref).abs().max() / ref.abs().max()).item()
                # SwiGLU on the interleaved layout
                act = torch.full((M, N // 2), float("nan"), device=dev, dtype=torch.bfloat16)
                C.wide_gemm(wp, N, K, x, act, 2, bm, bn, variant, 1, nst, None, 0.0, ws, True)
                g = ref.view(M, N // 16, 2, 8)
                want = (F.silu(g[:, :, 0]) * g[:, :, 1]).reshape(M, N // 2)
                e2 = ((act.float() - want).abs().max() / want.abs().max()).item()
                log.append({"check": True, "tile": [bm, bn], "variant": variant, "M": M, "N": N, "K": K, "fp8": fp8, "err_bf16": e0, "err_slab": e1, "err_swiglu": e2})
                worst = max(worst, e0, e1, e2)
                assert e0 < 1.5e-2 and e1 < 1e-4 and e2 < 1.5e-2, log[-1]
    print(f"correctness: worst relative error {worst:.2e} over {len(log)} cases", flush=True)


def bench(C, dev, rows, out, fp8, only=None):
    from kvmini.models.llama import quantize_fp8_rows

    shapes = {"qkv": (6144, 4096, 0), "o": (4096, 4096, 0), "gate_up": (28672, 4096, 2), "down": (4096, 14336, 0)}
    for name, (N, K, epi) in shapes.items():
        if only and name not in only:
            continue
        copies = max(2, int(600e6 // (N * K * 2)) + 1)
        Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(copies)]
        if fp8:
            qs = [quantize_fp8_rows(w) for w in Ws]
            Wps = [C.pack_weight_fp8(q.view(torch.uint8).contiguous()) for q, _ in qs]
            scales = [s.contiguous() for _, s in qs]
        else:
            Wps = [C.pack_weight(w) for w in Ws]
            scales = [None] * copies
        for M in rows:
            x = torch.randn(M, K, device=dev).bfloat16()
            st = {"i": 0}

            def nxt():
                st["i"] = (st["i"] + 1) % copies
                return st["i"]

            fns = {}
            if not fp8:
                def lib():
                    y = F.linear(x, Ws[nxt()])
                    if epi == 2:
                        a = torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev)
                        C.swiglu_interleaved(y, a)

                fns["library"] = lib
            n_out = N // 2 if epi == 2 else N
            y = torch.empty((M, n_out), dtype=torch.bfloat16, device=dev)
            vmap = VARIANTS_FP8 if fp8 else VARIANTS
            for bm, bn in TILES:
                if (bm, bn) not in vmap:
                    continue
                mt, ntl = -(-M // bm), N // bn
                if N % bn:
                    continue
                for variant in vmap[(bm, bn)]:
                    nst = K // (32 * C.wide_gemm_stage_ksteps(bm, bn, variant))
                    for mf in (True, False):
                        def mk(bm=bm, bn=bn, mf=mf, variant=variant, nst=nst):
                            def f():
                                i = nxt()
                                C.wide_gemm(Wps[i], N, K, x, y, epi, bm, bn, variant, 1, nst, None, 0.0, scales[i], mf)
                            return f
                        fns[f"wide_{bm}x{bn}_v{variant}_{'m' if mf else 'n'}"] = mk()
                for variant in ((0,) + tuple(v for v in vmap[(bm, bn)] if v >= 4)) if epi == 0 and M <= 512 else ():
                    nst = K // (32 * C.wide_gemm_stage_ksteps(bm, bn, variant))
                    for splits in (2, 4, 8):
                        if mt * ntl * splits > 1024 or mt * ntl >= 256:
                            continue
                        sps = -(-nst // splits)
                        if (splits - 1) * sps >= nst:
                            continue
                        sl = torch.empty((splits, M, N), dtype=torch.float32, device=dev)

                        def mk2(bm=bm, bn=bn, splits=splits, sps=sps, sl=sl, variant=variant):
                            def f():
                                i = nxt()
                                C.wide_gemm(Wps[i], N, K, x, sl, 1, bm, bn, variant, splits, sps, None, 0.0, scales[i], True)
                            return f
                        fns[f"wide_{bm}x{bn}_v{variant}_s{splits}"] = mk2()
            t = time_fns(fns)
            flop = 2.0 * M * N * K
            best = min((v[0], k) for k, v in t.items() if k != "library")
            rec = {"shape": name, "M": M, "N": N, "K": K, "fp8": fp8, "us": {k: round(v[0], 2) for k, v in t.items()}, "best": best[1], "best_us": round(best[0], 2),
                   "best_tflops": round(flop / best[0] / 1e6, 1), "library_tflops": round(flop / t["library"][0] / 1e6, 1) if "library" in t else None,
                   "weight_tb_s": round(N * K * (1 if fp8 else 2) / best[0] / 1e6, 2)}
            out.append(rec)
            print(json.dumps(rec), flush=True)
        del Ws, Wps
        torch.cuda.empty_cache()


def bench_fp8a(C, dev, rows, out, only=None):
    """FP8 weights x FP8 activations (scaled MFMA) against the same tiles with bf16 activations
    (FP8 weights widened in registers) and against the bf16 kernel's best tile; the per-token
    quantisation of the activations is timed on its own and is part of the fp8a totals."""
    from kvmini.models.llama import quantize_fp8_rows
    from kvmini.ops import hip_ops

    shapes = {"qkv": (6144, 4096, 0), "o": (4096, 4096, 0), "gate_up": (28672, 4096, 2), "down": (4096, 14336, 0)}
    for name, (N, K, epi) in shapes.items():
        if only and name not in only:
            continue
        copies = max(2, int(600e6 // (N * K)) + 1)
        Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(copies)]
        qs = [quantize_fp8_rows(w) for w in Ws]
        Wp8 = [C.pack_weight_fp8(q.view(torch.uint8).contiguous()) for q, _ in qs]
        scales = [s.contiguous() for _, s in qs]
        Wp16 = [C.pack_weight(w) for w in Ws[:2]]
        del Ws, qs
        for M in rows:
            x = torch.randn(M, K, device=dev).bfloat16()
            x8, a_scale = C.quantize_rows_fp8(x)
            n_out = N // 2 if epi == 2 else N
            y = torch.empty((M, n_out), dtype=torch.bfloat16, device=dev)
            st = {"i": 0}

            def nxt(n=copies):
                st["i"] = (st["i"] + 1) % n
                return st["i"]

            fns = {"quantize_rows": lambda: C.quantize_rows_fp8(x)}
            bm, bn, variant, _sp, sps = hip_ops._plan_wide(M, N, K, 0, False)
            fns["bf16_planned"] = lambda bm=bm, bn=bn, variant=variant, sps=sps: C.wide_gemm(Wp16[nxt(2)], N, K, x, y, epi, bm, bn, variant, 1, sps, None, 0.0, None, True)
            bm, bn, variant, _sp, sps = hip_ops._plan_wide(M, N, K, 1, False)
            fns["w8a16_planned"] = lambda bm=bm, bn=bn, variant=variant, sps=sps: (lambda i: C.wide_gemm(Wp8[i], N, K, x, y, epi, bm, bn, variant, 1, sps, None, 0.0, scales[i], True))(nxt())
            for (tbm, tbn), variants in VARIANTS_FP8.items():
                if N % tbn:
                    continue
                for variant in variants:
                    for mf in (True, False):
                        def mk(tbm=tbm, tbn=tbn, variant=variant, mf=mf):
                            def f():
                                i = nxt()
                                C.wide_gemm_fp8a(Wp8[i], N, K, x8, a_scale, y, epi, tbm, tbn, variant, 1, K // 128, None, 0.0, scales[i], mf)
                            return f
                        fns[f"fp8a_{tbm}x{tbn}_v{variant}_{'m' if mf else 'n'}"] = mk()
            t = time_fns(fns)
            flop = 2.0 * M * N * K
            best = min((v[0], k) for k, v in t.items() if k.startswith("fp8a_"))
            rec = {"bench": "fp8a", "shape": name, "M": M, "N": N, "K": K, "us": {k: round(v[0], 2) for k, v in t.items()}, "best": best[1], "best_us": round(best[0], 2),
                   "best_with_quantize_us": round(best[0] + t["quantize_rows"][0], 2), "best_tflops": round(flop / best[0] / 1e6, 1),
                   "bf16_tflops": round(flop / t["bf16_planned"][0] / 1e6, 1), "speedup_vs_bf16_incl_quantize": round(t["bf16_planned"][0] / (best[0] + t["quantize_rows"][0]), 3)}
            out.append(rec)
            print(json.dumps(rec), flush=True)
        del Wp8, Wp16
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fp8a", action="store_true", help="FP8 weights x FP8 activations (scaled MFMA) against bf16 / weight-only FP8")
    ap.add_argument("--out", default="gpurun_out/wide_gemm.json")
    ap.add_argument("--rows", default="256,512,1024,2048")
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--no-bench", action="store_true")
    ap.add_argument("--shapes", default="qkv,o,gate_up,down")
    a = ap.parse_args()
    from kvmini.ops import native

    C = native()
    dev = torch.device("cuda:0")
    out: list = []
    if a.fp8a:
        bench_fp8a(C, dev, [int(r) for r in a.rows.split(",")], out, a.shapes.split(","))
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
        return
    check(C, dev, out)
    if not a.no_bench:
        bench(C, dev, [int(r) for r in a.rows.split(",")], out, a.fp8, a.shapes.split(","))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
