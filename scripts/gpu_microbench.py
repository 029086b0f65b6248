#!/usr/bin/env python3
"""Kernel- and step-level timings on one MI355X (writes JSON under gpurun_out/).

* streaming GEMM variants vs the library GEMM for the Llama-3-8B decode shapes;
  reports achieved weight-streaming bandwidth (HBM roofline is ~6.3 TB/s achievable);
* one full decode step of the engine (HIP graph) at a given batch.
All variants are timed interleaved in one process (median of rounds).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def time_fns(fns, rounds=20, inner=10):
    """fns: dict name -> callable. Interleaved rounds; returns median us per call."""
    res = {k: [] for k in fns}
    for k, f in fns.items():
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
    return {k: statistics.median(v) for k, v in res.items()}


def gemm_bench(out):
    from kvmini.ops import native

    C = native()
    dev = torch.device("cuda:0")
    shapes = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (28672, 4096), "down": (4096, 14336), "lm_head": (128256, 4096)}
    # rotate through several weight copies so the 256 MiB Infinity Cache cannot serve re-reads
    for name, (N, K) in shapes.items():
        copies = max(2, int(600e6 // (N * K * 2)) + 1)
        Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(copies)]
        Wps = [C.pack_weight(w) for w in Ws]
        for M in (16, 32, 64):
            x = torch.randn(M, K, device=dev).bfloat16()
            y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            st = {"i": 0}

            def nxt(lst):
                st["i"] = (st["i"] + 1) % copies
                return lst[st["i"]]

            def sg(ks, slabs):
                def f():
                    wp = nxt(Wps)
                    if slabs:
                        splits, cps = C.stream_gemm_plan(N, K, ks, 16)
                        out = torch.empty((splits, M, N), dtype=torch.float32, device=dev)
                        C.stream_gemm(wp, N, K, x, out, 1, ks, splits, cps, None, 0.0)
                    else:
                        C.stream_gemm(wp, N, K, x, y, 0, ks, 1, (K // 32) // ks, None, 0.0)
                return f

            def sg3(nb, slabs, target=512):
                def f():
                    wp = nxt(Wps)
                    if slabs:
                        splits, gps = C.stream_gemm3_plan(N, K, nb, 32, target)
                        out = torch.empty((splits, M, N), dtype=torch.float32, device=dev)
                        C.stream_gemm3(wp, N, K, x, out, 1, nb, splits, gps, None, 0.0)
                    else:
                        C.stream_gemm3(wp, N, K, x, y, 0, nb, 1, K // (128 * nb), None, 0.0)
                return f

            v3 = {}
            for nb in (2, 4):
                if K % (128 * nb):
                    continue
                if N > 8192:
                    v3[f"v3_nb{nb}"] = sg3(nb, False)
                else:
                    for target in (512, 768):
                        v3[f"v3_nb{nb}_slab{target}"] = sg3(nb, True, target)
            fns = {
                **v3,
                "v2_ks4": sg(4, False),
                "v2_ks8": sg(8, False),
                "v2_ks4_slab": sg(4, True),
                "v2_ks8_slab": sg(8, True),
                "hipblaslt": lambda: torch.nn.functional.linear(x, nxt(Ws)),
            }
            t = time_fns(fns)
            row = {"shape": name, "N": N, "K": K, "M": M, "us": t, "TBps": {k: N * K * 2 / (v * 1e-6) / 1e12 for k, v in t.items()}}
            print(json.dumps(row), flush=True)
            out.append(row)
        del Ws, Wps
        torch.cuda.empty_cache()


def balance_bench(out):
    """Unsplit v3 GEMM (K = 4096, M = 32) over widths that give 1, 1.5, 1.75, 2, 2.5, 3 and 7.8
    workgroups per CU: is the gate/up projection (448 workgroups on 256 CUs) paying for the
    uneven last round?"""
    from kvmini.ops import native

    C = native()
    dev = torch.device("cuda:0")
    K, M = 4096, 32
    x = torch.randn(M, K, device=dev).bfloat16()
    for N in (16384, 24576, 28672, 32768, 40960, 49152, 128256):
        copies = max(3, int(800e6 // (N * K * 2)) + 1)
        Wps = [C.pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        st = {"i": 0}

        def mk(nb):
            def f():
                st["i"] = (st["i"] + 1) % copies
                C.stream_gemm3(Wps[st["i"]], N, K, x, y, 0, nb, 1, K // (128 * nb), None, 0.0)
            return f

        t = time_fns({"nb4": mk(4), "nb2": mk(2)})
        row = {"bench": "balance", "N": N, "wgs": N // 64, "wgs_per_cu": N / 64 / 256, "us": t, "TBps": {k: N * K * 2 / (v * 1e-6) / 1e12 for k, v in t.items()}}
        print(json.dumps(row), flush=True)
        out.append(row)
        del Wps
        torch.cuda.empty_cache()


def nw_bench(out, rows=(32, 64)):
    """Waves per workgroup of the v3 GEMM (4 / 6 / 7 / 8) on the decode shapes at M = 32 and 64
    (or, ``--what wide``, the 128 / 256-row variants against the library GEMM)."""
    from kvmini.ops import native

    C = native()
    dev = torch.device("cuda:0")
    shapes = {"qkv": (6144, 4096, True), "o": (4096, 4096, True), "gate_up": (28672, 4096, False), "down": (4096, 14336, True), "lm_head": (128256, 4096, False)}
    for name, (N, K, slabs) in shapes.items():
        copies = max(3, int(800e6 // (N * K * 2)) + 1)
        Wps = [C.pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
        Ws = None
        for M in rows:
            x = torch.randn(M, K, device=dev).bfloat16()
            y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            st = {"i": 0}

            def mk(nw, nb, target=512):
                def f():
                    st["i"] = (st["i"] + 1) % copies
                    if slabs:
                        splits, gps = C.stream_gemm3_plan(N, K, nb, 32, target, nw)
                        o = torch.empty((splits, M, N), dtype=torch.float32, device=dev)
                        C.stream_gemm3(Wps[st["i"]], N, K, x, o, 1, nb, splits, gps, None, 0.0, None, nw)
                    else:
                        C.stream_gemm3(Wps[st["i"]], N, K, x, y, 0, nb, 1, K // (128 * nb), None, 0.0, None, nw)
                return f

            fns = {}
            if M > 64:
                if Ws is None:
                    Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(copies)]

                def lib():
                    st["i"] = (st["i"] + 1) % copies
                    torch.nn.functional.linear(x, Ws[st["i"]])

                fns["library"] = lib
            for nw in (4, 6, 7, 8):
                if N % (16 * nw):
                    continue
                for nb in ((2,) if M > 64 else (2, 4)):
                    if slabs:
                        for target in (256, 512):
                            fns[f"nw{nw}_nb{nb}_t{target}"] = mk(nw, nb, target)
                    else:
                        fns[f"nw{nw}_nb{nb}"] = mk(nw, nb)
            t = time_fns(fns)
            row = {"bench": "nw", "shape": name, "N": N, "K": K, "M": M, "us": {k: round(v, 2) for k, v in t.items()}}
            print(json.dumps(row), flush=True)
            out.append(row)
        del Wps
        torch.cuda.empty_cache()


def attn_bench(out):
    """The tiled prompt-attention kernel alone: one 1024-query chunk of a Llama-3-8B layer
    (32 q heads / 8 kv heads x 128) at positions 0..1023 and 1024..2047, and a 2048-query chunk."""
    import math

    from kvmini.ops import hip_ops

    dev = torch.device("cuda:0")
    nh, nkv, hd, bs = int(os.environ.get("ATTN_HEADS", "32")), 8, 128, 16
    ctx = int(os.environ.get("ATTN_CTX", "4096"))  # longest position any shape reaches
    nb = ctx // bs + 8
    kc = (torch.randn(nb, nkv, bs, hd, device=dev) * 0.5).bfloat16()
    vc = (torch.randn(nb, nkv, bs, hd, device=dev) * 0.5).bfloat16()
    bt = torch.randperm(nb, device=dev)[: ctx // bs].view(1, -1).int().contiguous()
    shapes = ((0, 1024), (1024, 1024), (0, 2048), (2048, 2048))
    if os.environ.get("ATTN_SHAPES"):  # "pos0:queries,..." (a counter run wants one shape only)
        shapes = tuple(tuple(int(v) for v in sh.split(":")) for sh in os.environ["ATTN_SHAPES"].split(","))
    for p0, n in shapes:
        q = torch.randn(n, nh, hd, device=dev).bfloat16()
        o = torch.empty(n, nh * hd, device=dev, dtype=torch.bfloat16)
        tiles = torch.tensor([(off, 16, p0 + off, 0) for off in range(0, n, 16)], device=dev, dtype=torch.int32).T.contiguous()
        quads = os.environ.get("ATTN_QUADS", "0") == "1"  # the 8-wave kernel (groups of four tiles)
        t = time_fns({"tiled": lambda: hip_ops.prefill_attention(q, kc, vc, bt, tiles, o, 1 / math.sqrt(hd), quads=quads)}, rounds=10, inner=5)["tiled"]
        flops = 4.0 * hd * nh * sum(p0 + i + 1 for i in range(n))
        row = {"bench": "prefill_attention", "quads": quads, "heads": nh, "pos0": p0, "queries": n, "us": round(t, 1), "TFLOPs": round(flops / t / 1e6, 1)}
        print(json.dumps(row), flush=True)
        out.append(row)


def decode_attn_bench(out):
    """The paged decode-attention kernel alone: 20 single-token queries of a Llama-3-8B layer
    (32 q heads / 8 kv heads x 128), one per sequence, against contexts of 128 .. 8000 tokens;
    the K/V bytes read per call over the time."""
    import math

    from kvmini.ops import hip_ops

    dev = torch.device("cuda:0")
    nh, nkv, hd, bs, nseq = 32, 8, 128, 16, 20
    for ctx in (128, 1024, 4096, 8000):
        per = (ctx + bs - 1) // bs
        nb = nseq * per
        kc = (torch.randn(nb, nkv, bs, hd, device=dev) * 0.5).bfloat16()
        vc = (torch.randn(nb, nkv, bs, hd, device=dev) * 0.5).bfloat16()
        q = torch.randn(nseq, nh, hd, device=dev).bfloat16()
        tok_seq = torch.arange(nseq, device=dev, dtype=torch.int32)
        ctx_lens = torch.full((nseq,), ctx, device=dev, dtype=torch.int32)
        gb = 2.0 * nseq * ctx * nkv * hd * 2 / 1e9
        # block order: as a fresh engine hands blocks out (consecutive per sequence) and as a
        # long-running one does (any free block)
        for order, bt in (("sequential", torch.arange(nb, device=dev)), ("shuffled", torch.randperm(nb, device=dev))):
            bt = bt.view(nseq, per).int().contiguous()
            t = time_fns({"paged": lambda: hip_ops.paged_attention(q, kc, vc, bt, tok_seq, ctx_lens, 1 / math.sqrt(hd))}, rounds=30, inner=20)["paged"]
            row = {"bench": "paged_decode_attention", "seqs": nseq, "ctx": ctx, "blocks": order, "us": round(t, 1), "kv_GB": round(gb, 3), "TBps": round(gb / t * 1e3, 2)}
            print(json.dumps(row), flush=True)
            out.append(row)
        del kc, vc


def step_bench(out, model, batch, kernels_list, quantization=None):
    kernels_list = [k for k in kernels_list if k == "hip"]  # the torch ops are not graph-capturable
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    for kernels in kernels_list:
        eng = LLMEngine(EngineConfig(model=model, device="cuda:0", max_num_seqs=max(32, batch), max_batched_tokens=max(256, batch), max_model_len=512 if kernels == "torch" else 2048, kernels=kernels, kv_memory_fraction=0.2, quantization=quantization))
        eng.warmup()
        prompts = [[1] + list(range(100 + i, 113 + i)) for i in range(batch)]
        for rep in range(2):
            seqs = [eng.make_sequence(p, SamplingParams(max_tokens=64, ignore_eos=True)) for p in prompts]
            for s in seqs:
                eng.scheduler.add(s)
            eng.step()  # prefill step
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            while eng.scheduler.has_work():
                eng.step()
                n += 1
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / max(1, n)
        row = {"bench": "decode_step", "model": model, "batch": batch, "kernels": kernels, "quantization": quantization, "ms_per_step": dt * 1e3, "steps": n, "weight_GB": eng.weights.nbytes() / 1e9, "eff_TBps": eng.weights.nbytes() / dt / 1e12, "load_s": eng.load_s, "warm_s": eng.warm_s}
        print(json.dumps(row), flush=True)
        out.append(row)
        del eng
        torch.cuda.empty_cache()


def mall_bench(out):
    """Does touching a weight matrix ahead of time (reads allocate in the 256 MiB
    memory-side cache) make the GEMM that streams it next faster?  cold = rotating
    copies, warm = same copy back to back, prefetched = prefetch kernel then GEMM
    (GEMM time only), plus the prefetch kernel's own time at several widths."""
    from kvmini.ops import native

    C = native()
    dev = torch.device("cuda:0")
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    shapes = {"o": (4096, 4096), "qkv": (6144, 4096), "down": (4096, 14336), "gate_up": (28672, 4096)}
    M = 32
    for name, (N, K) in shapes.items():
        copies = max(3, int(800e6 // (N * K * 2)) + 1)
        Wps = [C.pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
        x = torch.randn(M, K, device=dev).bfloat16()
        ks = 4
        splits, cps = C.stream_gemm_plan(N, K, ks, 16)
        slab = torch.empty((splits, M, N), dtype=torch.float32, device=dev)

        def gemm(wp):
            C.stream_gemm(wp, N, K, x, slab, 1, ks, splits, cps)

        def timed(pre, n=40):
            ts = []
            for i in range(n):
                wp = Wps[i % copies]
                if pre:
                    pre(wp)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gemm(wp)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            return statistics.median(ts)

        row = {"bench": "mall", "shape": name, "MB": N * K * 2 / 1e6, "gemm_cold_us": timed(None), "gemm_warm_us": timed(lambda wp: gemm(wp))}
        for wgs in (64, 256, 1024):
            row[f"gemm_after_prefetch{wgs}_us"] = timed(lambda wp: C.prefetch(wp, wgs, sink))
            ts = []
            for i in range(20):
                wp = Wps[i % copies]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                C.prefetch(wp, wgs, sink)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            row[f"prefetch{wgs}_us"] = statistics.median(ts)
        print(json.dumps(row), flush=True)
        out.append(row)
        del Wps
        torch.cuda.empty_cache()


def prefill_bench(out, model):
    """One engine step that prefills 256 prompt tokens (254 for the headline's cycle) and a
    2048-token prompt end to end, per step token budget, bf16 and FP8 weights."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    modes = ((None, 256), (None, 1024), (None, 2048), ("fp8", 1024), ("fp8", 2048), ("mxfp4", 2048))
    if os.environ.get("PREFILL_MODES"):  # "mode:budget,..." with mode in bf16 / fp8 / mxfp4
        modes = tuple((None if m == "bf16" else m, int(b)) for m, b in (x.split(":") for x in os.environ["PREFILL_MODES"].split(",")))
    for mode, budget in modes:
        eng = LLMEngine(EngineConfig(model=model, device="cuda:0", max_num_seqs=32, max_batched_tokens=budget, max_model_len=4096, kernels="hip", kv_memory_fraction=0.2, quantization=mode))
        eng.warmup()
        res = {}
        for plen in (254, 256, 2048):
            ts = []
            for rep in range(3):
                seq = eng.make_sequence([1] + [(7 * i + rep) % 1000 + 10 for i in range(plen - 1)], SamplingParams(max_tokens=1, ignore_eos=True))
                eng.scheduler.add(seq)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                while eng.scheduler.has_work():
                    eng.step()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res[plen] = min(ts) * 1e3
        row = {"bench": "prefill", "model": model, "quantization": mode, "max_batched_tokens": budget, "ms_254_tokens": res[254], "ms_256_tokens": res[256], "ms_2048_tokens": res[2048], "tok_per_s_2048": 2048 / res[2048] * 1e3, "mem_GB": torch.cuda.memory_allocated() / 1e9}
        print(json.dumps(row), flush=True)
        out.append(row)
        del eng
        torch.cuda.empty_cache()


def longctx_bench(out, model, lens=(8192, 32768, 131072), kv_dtypes=("auto", "fp8"), batch=1):
    """The long-context axis (runners/profiles/long-context.yaml): ``batch`` prompts of N tokens
    through the engine — prefill tokens/s, then 32 decode steps at that context (ms per step, host
    staging time per step), and the KV bytes the sequences hold; bf16 and e4m3 KV cache."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    for kv in kv_dtypes:
        for plen in lens:
            # (several prompts: the first ones decode while the later ones are fed, so every sequence gets enough
            # tokens to still be running when the last prompt is in; the decode phase is then timed over 32 steps)
            extra = 0 if batch == 1 else batch * (plen // 2048 + 2)
            eng = LLMEngine(EngineConfig(model=model, device="cuda:0", max_num_seqs=max(2, batch), max_batched_tokens=2048, max_model_len=plen + 64 + extra, kernels="hip", kv_memory_fraction=0.6,
                                         kv_cache_dtype=kv, enable_prefix_caching=False))
            eng.warmup()
            seqs = [eng.make_sequence([1] + [(7 * i + 13 * b) % 5000 + 10 for i in range(plen - 1)], SamplingParams(max_tokens=33 + extra, ignore_eos=True)) for b in range(batch)]
            for s_ in seqs:
                assert eng.scheduler.add(s_) is None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            while any(s_.first_token_time is None for s_ in seqs):
                eng.step()
            torch.cuda.synchronize()
            t_prefill = time.perf_counter() - t0
            n = 0
            t0 = time.perf_counter()
            while eng.scheduler.has_work() and n < 32:
                eng.step()
                n += 1
            torch.cuda.synchronize()
            t_dec = (time.perf_counter() - t0) / max(1, n)
            running = len(eng.scheduler.running)
            r = eng.runner
            per_tok = r.cfg.num_layers * 2 * r.nkv * r.hd * r.kv_cache.element_size()
            # host side of one decode step: staging the image (block tables included) without running it
            batch_obj = None
            row = {"bench": "long_context", "model": model, "kv_cache_dtype": kv, "prompt_tokens": plen, "batch": batch, "prefill_s": round(t_prefill, 3), "prefill_tok_per_s": round(batch * plen / t_prefill),
                   "decode_ms_per_step": round(t_dec * 1e3, 3), "decode_steps": n, "sequences_decoding": running, "kv_GB_held": round(batch * plen * per_tok / 1e9, 3), "kv_bytes_per_token": per_tok,
                   "kv_blocks_total": r.num_blocks, "stage_ints": int(r.stage_dev.numel()), "mem_GB": round(torch.cuda.memory_allocated() / 1e9, 1)}
            print(json.dumps(row), flush=True)
            out.append(row)
            del eng, seqs, batch_obj
            torch.cuda.empty_cache()


def ridealong_bench(out, model, lens=(32768, 131072), kv_dtypes=("auto",)):
    """A sequence of N tokens keeps decoding while a second N-token prompt is fed in 2048-token
    chunks: time of the steps that carry both (the first sequence's inter-token time during the
    other's prefill) and the second prompt's prefill rate — with the decode rows on the split-KV
    kernel, and (threshold out of reach) on the per-token kernel."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    for kv in kv_dtypes:
        for plen in lens:
            for split in (True, False):
                eng = LLMEngine(EngineConfig(model=model, device="cuda:0", max_num_seqs=2, max_batched_tokens=2048, max_model_len=plen + 4096, kernels="hip", kv_memory_fraction=0.6,
                                             kv_cache_dtype=kv, enable_prefix_caching=False))
                if not split:
                    eng.runner.DEC_SPLIT_MIN_CTX = 1 << 30
                eng.warmup()
                s1 = eng.make_sequence([1] + [(7 * i) % 5000 + 10 for i in range(plen - 1)], SamplingParams(max_tokens=2048, ignore_eos=True))
                assert eng.scheduler.add(s1) is None
                while s1.first_token_time is None:
                    eng.step()
                for _ in range(4):
                    eng.step()
                s2 = eng.make_sequence([1] + [(11 * i) % 5000 + 10 for i in range(plen - 1)], SamplingParams(max_tokens=2, ignore_eos=True))
                assert eng.scheduler.add(s2) is None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 0
                while s2.first_token_time is None:
                    eng.step()
                    n += 1
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                row = {"bench": "decode_rides_along_with_prefill", "model": model, "kv_cache_dtype": kv, "context_tokens": plen, "decode_rows_on_split_kv": split, "steps": n,
                       "ms_per_step": round(dt / max(1, n) * 1e3, 2), "second_prompt_prefill_s": round(dt, 3), "second_prompt_tok_per_s": round(plen / dt)}
                print(json.dumps(row), flush=True)
                out.append(row)
                del eng, s1, s2
                torch.cuda.empty_cache()


def deepspec_bench(out, model, lens=(32768,), kv_dtypes=("auto",)):
    """Speculative decoding (self-drafting, 4 layers, 3 proposals) N tokens into a sequence: ms per
    verification step with the chunk's rows on the split-KV kernel and, threshold out of reach,
    as one tile of the prompt attention kernel."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    for kv in kv_dtypes:
        for plen in lens:
            for split in (True, False):
                eng = LLMEngine(EngineConfig(model=model, device="cuda:0", max_num_seqs=2, max_batched_tokens=2048, max_model_len=plen + 4096, kernels="hip", kv_memory_fraction=0.5,
                                             kv_cache_dtype=kv, enable_prefix_caching=False, draft_model="self:4", num_speculative_tokens=3))
                if not split:
                    eng.runner.DEC_SPLIT_MIN_CTX = 1 << 30
                eng.warmup()
                s1 = eng.make_sequence([1] + [(7 * i) % 5000 + 10 for i in range(plen - 1)], SamplingParams(max_tokens=64, ignore_eos=True))
                assert eng.scheduler.add(s1) is None
                while s1.first_token_time is None:
                    eng.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 0
                while eng.scheduler.has_work():
                    eng.step()
                    n += 1
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                row = {"bench": "speculative_verification_deep_in_context", "model": model, "kv_cache_dtype": kv, "context_tokens": plen, "rows_on_split_kv": split, "steps": n,
                       "ms_per_step": round(dt / max(1, n) * 1e3, 2), "tokens": len(s1.output_ids), "ms_per_token": round(dt / max(1, len(s1.output_ids) - 1) * 1e3, 2)}
                print(json.dumps(row), flush=True)
                out.append(row)
                del eng, s1
                torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="gemm,step")
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--batch", type=int, default=20)
    ap.add_argument("--kernels", default="hip,torch")
    ap.add_argument("--out", default="gpurun_out/microbench.json")
    ap.add_argument("--quantization", default=None)
    ap.add_argument("--lens", default="8192,32768,131072")
    ap.add_argument("--kv-dtypes", default="auto,fp8")
    a = ap.parse_args()
    out = []
    if "gemm" in a.what:
        gemm_bench(out)
    if "mall" in a.what:
        mall_bench(out)
    if "balance" in a.what:
        balance_bench(out)
    if "attn" in a.what.split(","):
        attn_bench(out)
    if "decode_attn" in a.what.split(","):
        decode_attn_bench(out)
    if "nw" in a.what.split(","):
        nw_bench(out)
    if "wide" in a.what.split(","):
        nw_bench(out, rows=(128, 256))
    if "prefill" in a.what.split(","):
        prefill_bench(out, a.model)
    if "deepspec" in a.what.split(","):
        deepspec_bench(out, a.model, lens=tuple(int(x) for x in a.lens.split(",")), kv_dtypes=tuple(a.kv_dtypes.split(",")))
    if "ridealong" in a.what.split(","):
        ridealong_bench(out, a.model, lens=tuple(int(x) for x in a.lens.split(",")), kv_dtypes=tuple(a.kv_dtypes.split(",")))
    if "longctx" in a.what.split(","):
        longctx_bench(out, a.model, lens=tuple(int(x) for x in a.lens.split(",")), kv_dtypes=tuple(a.kv_dtypes.split(",")), batch=a.batch if a.batch != 20 else 1)
    if "step" in a.what:
        step_bench(out, a.model, a.batch, a.kernels.split(","), a.quantization)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
