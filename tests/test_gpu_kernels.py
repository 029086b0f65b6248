"""Numerics of the gfx950 HIP kernels against plain PyTorch / numpy fp32-fp64 references."""

import math
import random
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def C():
    from kvmini.ops import native

    return native()


def _dev():
    return torch.device("cuda:0")


def test_add_rmsnorm(C):
    from kvmini.ops import torch_ops

    torch.manual_seed(2)
    T, H = 37, 4096
    x = torch.randn(T, H, device=_dev()).bfloat16()
    r = torch.randn(T, H, device=_dev()).bfloat16()
    w = (1 + 0.1 * torch.randn(H, device=_dev())).bfloat16()
    y_ref, r_ref = torch_ops.add_rmsnorm(x, r.clone(), w, 1e-5)
    y = torch.empty_like(x)
    r2 = r.clone()
    C.add_rmsnorm(x, r2, w, y, 1e-5)
    assert torch.equal(r2, r_ref)
    assert (y.float() - y_ref.float()).abs().max().item() <= 2e-2


def _rand_cache(nb, nkv, bs, hd):
    return torch.randn(nb, nkv, bs, hd, device=_dev()).bfloat16()


def test_rope_kv_write(C):
    from kvmini.models.llama import get_config, rope_tables
    from kvmini.ops import torch_ops

    torch.manual_seed(3)
    nh, nkv, hd, bs, nb, T = 8, 2, 128, 16, 32, 21
    cfg = get_config("llama-3-8b")
    cos, sin = rope_tables(cfg, 512, _dev())
    qkv = torch.randn(T, (nh + 2 * nkv) * hd, device=_dev()).bfloat16()
    pos = torch.randint(0, 512, (T,), device=_dev(), dtype=torch.int32)
    slots = torch.randperm(nb * bs, device=_dev())[:T].int()
    slots[5] = -1
    kc, vc = _rand_cache(nb, nkv, bs, hd), _rand_cache(nb, nkv, bs, hd)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    q_ref = torch_ops.rope_kv_write(qkv.clone(), pos.long(), cos, sin, kc_ref, vc_ref, slots.long(), nh, nkv, hd)
    q_in = qkv.clone()
    C.rope_kv_write(q_in, pos, cos, sin, kc, vc, slots, nh, nkv, hd)
    q = q_in[:, : nh * hd].view(T, nh, hd)
    assert (q.float() - q_ref.float()).abs().max().item() <= 3e-2
    assert (kc.float() - kc_ref.float()).abs().max().item() <= 3e-2
    assert torch.equal(vc, vc_ref)


@pytest.mark.parametrize("rep", [1, 3, 4, 7, 8])
@pytest.mark.parametrize("hd", [128, 64])
def test_paged_attention(C, rep, hd):
    from kvmini.ops import torch_ops

    torch.manual_seed(4)
    nkv, bs, nb = 2, 16, 64
    nh = nkv * rep
    n_seqs, max_blocks = 4, 12
    T = 9
    kc, vc = _rand_cache(nb, nkv, bs, hd), _rand_cache(nb, nkv, bs, hd)
    bt = torch.randperm(nb, device=_dev())[: n_seqs * max_blocks].view(n_seqs, max_blocks).int()
    tok_seq = torch.randint(0, n_seqs, (T,), device=_dev(), dtype=torch.int32)
    ctx = torch.tensor([1, 5, 16, 17, 33, 100, 190, 0, 64], device=_dev(), dtype=torch.int32)
    q = torch.randn(T, nh, hd, device=_dev()).bfloat16()
    scale = 1 / math.sqrt(hd)
    ref = torch_ops.paged_attention(q, kc, vc, bt, tok_seq, ctx, scale)
    out = torch.empty(T, nh * hd, device=_dev(), dtype=torch.bfloat16)
    C.paged_attention(q, kc, vc, bt, tok_seq, ctx, out, scale)
    assert (out.float() - ref.float()).abs().max().item() <= 2e-2
    assert out[7].abs().max().item() == 0  # padding token


def test_argmax(C):
    torch.manual_seed(5)
    logits = torch.randn(19, 128256, device=_dev()).bfloat16()
    out = torch.empty(19, dtype=torch.int64, device=_dev())
    C.argmax(logits, out)
    assert torch.equal(out, logits.float().argmax(-1))


def test_swiglu_interleaved(C):
    from kvmini.ops import torch_ops

    gu = torch.randn(7, 2048, device=_dev()).bfloat16()
    act = torch.empty(7, 1024, device=_dev(), dtype=torch.bfloat16)
    C.swiglu_interleaved(gu, act)
    ref = torch_ops.swiglu_interleaved(gu)
    assert (act.float() - ref.float()).abs().max().item() <= 2e-2


# ---- latency statistics ------------------------------------------------------------
def test_quantiles_golden():
    from kvmini.ops import hist_quantile as hq

    v = [10, 20, 30, 40, 50, 60, 70, 80, 90, 100]
    p50, p95 = hq.quantiles(v, (0.5, 0.95))
    assert p50 == 55 and abs(p95 - 95.5) < 1e-12


@pytest.mark.parametrize("n", [1, 2, 1000, 100003, 3_000_000])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_quantiles_exact_vs_numpy(n, dtype):
    from kvmini.analysis.quantiles import percentiles
    from kvmini.ops import hist_quantile as hq

    rng = np.random.default_rng(n)
    a = rng.lognormal(3.0, 0.7, n).astype(dtype)
    if n > 10:
        a[rng.integers(0, n, n // 50)] = np.nan
    got = hq.batched_quantiles(a.reshape(1, -1), (0.5, 0.95, 0.99))[0]
    want = percentiles(a.astype(np.float64), (0.5, 0.95, 0.99), engine="numpy")
    assert np.allclose(got, want, rtol=0, atol=0), (got, want)


def test_quantiles_batched_ragged_and_degenerate():
    from kvmini.analysis.quantiles import percentiles
    from kvmini.ops import hist_quantile as hq

    rng = np.random.default_rng(7)
    n = 50000
    mat = np.full((4, n), np.nan)
    mat[0, :] = rng.exponential(100, n)
    mat[1, :1234] = rng.normal(0, 5, 1234)  # negative values too
    mat[2, :] = 42.0  # all identical
    got, stats = hq.batched_quantiles(mat, (0.5, 0.95, 0.99), return_stats=True)
    for i in range(3):
        want = percentiles(mat[i], (0.5, 0.95, 0.99), engine="numpy")
        assert np.array_equal(got[i], np.asarray(want)), (i, got[i], want)
    assert np.isnan(got[3]).all() and stats[3]["count"] == 0
    assert stats[0]["count"] == n and abs(stats[0]["mean"] - mat[0].mean()) < 1e-6
    assert stats[1]["min"] == mat[1, :1234].min() and stats[1]["max"] == mat[1, :1234].max()


def test_equal_width_histogram_matches_cpu():
    from kvmini.analysis.histograms import compute_histogram
    from kvmini.ops import hist_quantile as hq

    rng = np.random.default_rng(11)
    a = rng.lognormal(2.0, 0.5, 200000)
    g = hq.equal_width_histogram(a, 20)
    c = compute_histogram(a.tolist(), 20, engine="python")
    assert g["counts"] == c["counts"] and sum(g["counts"]) == a.size
    assert np.allclose(g["bin_edges"], c["bin_edges"])


def test_streaming_histogram_atomics_and_quantiles():
    from kvmini.ops import hist_quantile as hq

    rng = np.random.default_rng(13)
    sh = hq.StreamingHistogram(1)
    chunks = [rng.lognormal(1.0, 0.4, 250000).astype(np.float32) for _ in range(4)]
    for c in chunks:
        sh.update(c)
    allv = np.concatenate(chunks)
    snap = sh.snapshot((0.5, 0.95, 0.99))[0]
    assert snap["count"] == allv.size  # no lost LDS/global atomic updates
    assert int(sh.hist.sum().item()) == allv.size
    for p, v in snap["quantiles"].items():
        assert abs(v - np.quantile(allv, p)) / np.quantile(allv, p) < 0.02


def test_energy_trapz(C):
    from kvmini.telemetry.energy import PowerSample, trapezoidal_wh

    rng = np.random.default_rng(17)
    G, S = 8, 400
    ts = np.sort(rng.uniform(1000, 1100, (G, S)), axis=1)
    w = rng.uniform(300, 1200, (G, S))
    w[2, 10:20] = np.nan
    n = np.full(G, S, dtype=np.int32)
    n[5] = 100
    out = C.energy_trapz(torch.from_numpy(ts).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(n).cuda(), 1010.0, 1090.0).cpu().numpy()
    for g in range(G):
        samples = [PowerSample(float(ts[g, i]), None if np.isnan(w[g, i]) else float(w[g, i])) for i in range(n[g])]
        assert abs(out[g, 0] - trapezoidal_wh(samples, 1010.0, 1090.0)) < 1e-9


# ---- engine end to end ---------------------------------------------------------------
@pytest.mark.parametrize("graphs,model", [(False, "small-hd128"), (True, "small-hd128"), (False, "small-hd128-qkv-bias"), (True, "small-hd128-qkv-bias")])
def test_engine_hip_matches_dense_reference(graphs, model):
    """(the -qkv-bias model: Qwen2 form — the bias rides in the first split-K slab, or is added to the bf16 output)"""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights
    from kvmini.models.reference import dense_logits

    w = random_weights(get_config(model), "cuda:0", torch.bfloat16, seed=3)
    oracle = random_weights(get_config(model), "cuda:0", torch.bfloat16, seed=3)  # the engine packs its copy in place
    for ww in (w, oracle):
        for lw in ww.layers:
            if lw.bqkv is not None:
                lw.bqkv.mul_(20.0)  # (N(0, 0.02) draws: make the bias matter)
    eng = LLMEngine(EngineConfig(model=model, device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, num_blocks=128, use_graphs=graphs, kernels="hip"), weights=w)
    if graphs:
        eng.warmup()
        assert eng.runner.graphs
    prompts = [[1, 50, 60, 70, 80], list(range(1, 41)), [1, 9, 8, 7, 6, 5, 4, 3, 2]]
    outs = eng.generate(prompts, SamplingParams(max_tokens=8, ignore_eos=True))
    # bf16 engine vs fp32 dense oracle: every generated token must be (near-)top of the oracle's logits
    for p, o in zip(prompts, outs):
        ids = list(p)
        for tok in o:
            lg = dense_logits(oracle, ids)[-1]
            top = lg.max().item()
            assert lg[tok].item() >= top - 0.05 * abs(top) - 0.05, (p, o)
            ids.append(tok)


# ---- weight prefetch rows of the residual / decode attention kernels ---------------------
def test_residual_reduce_prefetch_rows_do_not_change_the_result(C):
    """The extra (prefetch) workgroups of residual_reduce only read weights: residual and sums
    of squares equal the plain launch bit for bit, for slab and bf16 inputs; plans whose ranges
    would leave the weight allocation are refused on the host."""
    from kvmini.ops import hip_ops

    torch.manual_seed(5)
    T, H = 32, 4096
    w = hip_ops.PackedLinear((torch.randn(6144, 4096, device=_dev()) * 0.02).bfloat16())
    hip_ops._prefetch_config.cache_clear()
    os.environ["KVMINI_PREFETCH"] = "qkv=16:1536,gu=12:1792"
    try:
        plan = hip_ops.prefetch_plan(w, T, "qkv")
        gu_plan = hip_ops.prefetch_plan(w, T, "gu")
        assert hip_ops.prefetch_plan(w, T, "o") is None  # site not configured
        assert hip_ops.prefetch_plan(w, 20, "qkv") is None  # not a whole number of 16-row tiles
    finally:
        del os.environ["KVMINI_PREFETCH"]
        hip_ops._prefetch_config.cache_clear()
    assert hip_ops.prefetch_plan(w, T, "qkv") is None  # off by default
    assert plan is not None and gu_plan is not None
    gx, gy, nw, kib, tile_bytes, split_bytes, wgs = plan[1]
    assert gx * nw * 16 == 6144 and tile_bytes == 4096 * 32 and split_bytes * gy == tile_bytes and wgs % 8 == 0
    slabs = torch.randn(4, T, H, device=_dev())
    xb = torch.randn(T, H, device=_dev()).bfloat16()
    for src in (slabs, xb):
        for pf in (plan, gu_plan):
            r0 = torch.randn(T, H, device=_dev()).bfloat16()
            r1 = r0.clone()
            ss0 = C.residual_reduce(src, r0, True)
            ss1 = C.residual_reduce(src, r1, True, *pf)
            torch.cuda.synchronize()
            assert torch.equal(r0, r1) and torch.equal(ss0, ss1)
    bad = list(plan[1])
    bad[3] = split_bytes // 1024 + 1  # more KiB than a range holds
    with pytest.raises(RuntimeError):
        C.residual_reduce(xb, xb.clone(), True, plan[0], bad, plan[2])
    bad = list(plan[1])
    bad[0] = gx + 1  # one feature group more than the weights have
    with pytest.raises(RuntimeError):
        C.residual_reduce(xb, xb.clone(), True, plan[0], bad, plan[2])
    bad = list(plan[1])
    bad[6] = 12  # not a multiple of 8
    with pytest.raises(RuntimeError):
        C.residual_reduce(xb, xb.clone(), True, plan[0], bad, plan[2])


@pytest.mark.parametrize("graphs", [False, True])
def test_engine_with_weight_prefetch_generates_the_same_tokens(graphs):
    """KVMINI_PREFETCH on (all four sites, incl. the rows of the fused decode attention launch):
    same tokens as with it off."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights
    from kvmini.ops import hip_ops

    prompts = [[1, 50, 60, 70, 80], list(range(1, 41)), [1, 9, 8, 7, 6, 5, 4, 3, 2]]
    outs = {}
    for spec in ("0", "qkv=2:64,o=2:16,gu=2:64,head=2:64"):
        hip_ops._prefetch_config.cache_clear()
        os.environ["KVMINI_PREFETCH"] = spec
        try:
            w = random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=3)
            eng = LLMEngine(EngineConfig(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, num_blocks=128, use_graphs=graphs, kernels="hip"), weights=w)
            if graphs:
                eng.warmup()
            if spec != "0":
                lw = eng.weights.layers[0]
                assert hip_ops.prefetch_plan(lw.wo, 16, "o") is not None and hip_ops.prefetch_plan(lw.wqkv, 16, "qkv") is not None
            outs[spec] = eng.generate(prompts, SamplingParams(max_tokens=8, ignore_eos=True))
            del eng
        finally:
            del os.environ["KVMINI_PREFETCH"]
            hip_ops._prefetch_config.cache_clear()
    a, b = outs.values()
    assert a == b


# ---- weight-streaming GEMM v2 ---------------------------------------------------------
@pytest.mark.parametrize("M", [16, 32, 48, 64, 20, 100, 70, 96])
@pytest.mark.parametrize("N,K", [(256, 128), (4096, 4096), (6144, 4096), (4096, 14336), (1000, 512)])
def test_stream_gemm_bf16_and_slabs(M, N, K):
    from kvmini.ops import hip_ops

    torch.manual_seed(0)
    W = (torch.randn(N, K, device=_dev()) * 0.05).bfloat16()
    x = torch.randn(M, K, device=_dev()).bfloat16()
    x[:, 0] += torch.arange(M, device=_dev()).bfloat16()
    ref = x.float() @ W.float().T
    tol = 2e-2 * ref.abs().max().item() + 1e-2
    pw = hip_ops.PackedLinear(W)
    y = hip_ops.linear(x, pw)
    assert y.shape == (M, N)
    assert (y.float() - ref).abs().max().item() <= tol
    s = hip_ops.linear(x, pw, allow_slabs=True)
    if isinstance(s, hip_ops.Partial):
        assert (s.slabs.sum(0) - ref).abs().max().item() <= tol
    else:
        assert (s.float() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("N,K,epi", [(22016, 4096, 2), (2368 * 16, 1024, 2), (1376 * 16, 512, 0), (208, 512, 0)])
@pytest.mark.parametrize("M", [32, 64])
def test_stream_gemm_ragged_last_workgroup(C, N, K, epi, M):
    """Widths whose 16-feature tiles do not divide by the waves of a workgroup (Llama-2-7B's and Qwen2.5-7B's gate/up
    projections): the last workgroup is ragged — its surplus waves stream the last tile again and store nothing —
    for every workgroup size, and the planner's pick goes through the op layer."""
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(2)
    W = (torch.randn(N, K, device=_dev()) * 0.05).bfloat16()
    x = torch.randn(M, K, device=_dev()).bfloat16()
    full = x.float() @ W.float().T
    ref = torch_ops.swiglu_interleaved(full) if epi == 2 else full
    tol = 2e-2 * ref.abs().max().item() + 1e-2
    pw = hip_ops.PackedLinear(W)
    for nw in (4, 6, 7, 8):
        n_out = N // 2 if epi == 2 else N
        out = torch.full((M, pw.n_pad // 2 if epi == 2 else pw.n_pad), float("nan"), dtype=torch.bfloat16, device=_dev())  # (widths are padded to whole tiles when packed)
        C.stream_gemm3(pw.wp, pw.n_pad, K, x, out, epi, 2, 1, K // 256, None, 0.0, None, nw)
        assert torch.isfinite(out.float()).all() and (out[:, :n_out].float() - ref.float()).abs().max().item() <= tol, nw
    y = hip_ops.linear_swiglu(x, pw) if epi == 2 else hip_ops.linear(x, pw)
    assert (y.float() - ref.float()).abs().max().item() <= tol
    assert hip_ops._waves_per_workgroup(22016) == 6 and hip_ops._waves_per_workgroup(28672) == 7


@pytest.mark.parametrize("nb", [2, 4])
@pytest.mark.parametrize("M", [16, 48])
def test_stream_gemm3_every_group_depth(nb, M):
    """The deep-pipeline kernel at both ring depths, unsplit and split-K, against fp32."""
    from kvmini.ops import native

    C = native()
    torch.manual_seed(4)
    N, K = 512, 2048
    W = (torch.randn(N, K, device=_dev()) * 0.05).bfloat16()
    x = torch.randn(M, K, device=_dev()).bfloat16()
    ref = x.float() @ W.float().T
    tol = 2e-2 * ref.abs().max().item() + 1e-2
    wp = C.pack_weight(W)
    y = torch.empty(M, N, device=_dev(), dtype=torch.bfloat16)
    C.stream_gemm3(wp, N, K, x, y, 0, nb, 1, K // (128 * nb))
    assert (y.float() - ref).abs().max().item() <= tol
    splits, gps = C.stream_gemm3_plan(N, K, nb, 16, 512)
    assert splits > 1 and (splits - 1) * gps < K // (128 * nb) <= splits * gps
    slabs = torch.empty(splits, M, N, device=_dev())
    C.stream_gemm3(wp, N, K, x, slabs, 1, nb, splits, gps)
    assert (slabs.sum(0) - ref).abs().max().item() <= tol


@pytest.mark.parametrize("M", [16, 64, 33])
def test_stream_gemm_swiglu(M):
    from kvmini.models.llama import interleave_gate_up
    from kvmini.ops import hip_ops

    torch.manual_seed(1)
    I, K = 1024, 512
    wg = (torch.randn(I, K, device=_dev()) * 0.05).bfloat16()
    wu = (torch.randn(I, K, device=_dev()) * 0.05).bfloat16()
    x = torch.randn(M, K, device=_dev()).bfloat16()
    act = hip_ops.linear_swiglu(x, hip_ops.PackedLinear(interleave_gate_up(wg, wu).contiguous()))
    ref = torch.nn.functional.silu(x.float() @ wg.float().T) * (x.float() @ wu.float().T)
    assert act.shape == (M, I)
    assert (act.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item() + 1e-2


def test_folded_norm_pipeline_matches_reference():
    """residual_reduce (slab sum + residual add + sum of squares) followed by projections
    that apply 1/rms in their epilogue == rmsnorm(residual + x) * gain @ W^T."""
    from kvmini.models.llama import get_config, interleave_gate_up, rope_tables
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(2)
    T, H, N = 32, 4096, 1024
    slabs = torch.randn(4, T, H, device=_dev())
    r = torch.randn(T, H, device=_dev()).bfloat16()
    gain = (1 + 0.1 * torch.randn(H, device=_dev())).bfloat16()
    W = (torch.randn(N, H, device=_dev()) * 0.03).bfloat16()
    y_norm, r_ref = torch_ops.add_rmsnorm(slabs.sum(0).bfloat16(), r.clone(), gain, 1e-5)
    ref = y_norm.float() @ W.float().T

    r2 = r.clone()
    h, r_new = hip_ops.add_rmsnorm(hip_ops.Partial(slabs), r2, gain, 1e-5)
    assert r_new is r2 and torch.equal(r2, r_ref)
    assert torch.allclose(h.ss.sum(-1), r_ref.float().pow(2).sum(-1), rtol=1e-4)
    Wf = hip_ops.PackedLinear((W.float() * gain.float()[None, :]).bfloat16())
    y = hip_ops.linear(h, Wf)
    tol = 3e-2 * ref.abs().max().item() + 2e-2
    assert (y.float() - ref).abs().max().item() <= tol
    # split-K slabs carry the row scale to their consumer
    p = hip_ops.linear(h, hip_ops.PackedLinear((torch.cat([W] * 4).float() * gain.float()[None, :]).bfloat16()), allow_slabs=True)
    assert isinstance(p, hip_ops.Partial) and p.ss is not None
    assert (p.reduce().float()[:, :N] - ref).abs().max().item() <= tol
    # first layer: no residual yet
    x0 = torch.randn(T, H, device=_dev()).bfloat16()
    h0, r0 = hip_ops.add_rmsnorm(x0, None, gain, 1e-5)
    assert torch.equal(r0, x0)
    # SwiGLU epilogue with the folded scale
    I = 512
    wg = (torch.randn(I, H, device=_dev()) * 0.03).bfloat16()
    wu = (torch.randn(I, H, device=_dev()) * 0.03).bfloat16()
    wgu = interleave_gate_up((wg.float() * gain.float()[None, :]).bfloat16(), (wu.float() * gain.float()[None, :]).bfloat16()).contiguous()
    act = hip_ops.linear_swiglu(h, hip_ops.PackedLinear(wgu))
    act_ref = torch.nn.functional.silu(y_norm.float() @ wg.float().T) * (y_norm.float() @ wu.float().T)
    assert (act.float() - act_ref).abs().max().item() <= 3e-2 * act_ref.abs().max().item() + 2e-2

    # RoPE consuming slabs (+ row scale)
    nh, nkv, hd, bs, nb = 8, 2, 128, 16, 32
    cos, sin = rope_tables(get_config("llama-3-8b"), 512, _dev())
    qs = torch.randn(3, T, (nh + 2 * nkv) * hd, device=_dev())
    pos = torch.randint(0, 512, (T,), device=_dev(), dtype=torch.int32)
    slots = torch.randperm(nb * bs, device=_dev())[:T].int()
    for ss in (None, h.ss):
        kc, vc = _rand_cache(nb, nkv, bs, hd), _rand_cache(nb, nkv, bs, hd)
        kc_ref, vc_ref = kc.clone(), vc.clone()
        part = hip_ops.Partial(qs, ss, 1e-5, H)
        q_ref = torch_ops.rope_kv_write(part.reduce(), pos.long(), cos, sin, kc_ref, vc_ref, slots.long(), nh, nkv, hd)
        q = hip_ops.rope_kv_write(part, pos, cos, sin, kc, vc, slots, nh, nkv, hd)
        assert (q.float() - q_ref.float()).abs().max().item() <= 6e-2
        assert (kc.float() - kc_ref.float()).abs().max().item() <= 6e-2
        assert (vc.float() - vc_ref.float()).abs().max().item() <= 6e-2


# ---- FP8 (OCP e4m3) weight-only path of the streaming GEMM ---------------------------
@pytest.mark.parametrize("M", [16, 32, 64, 20, 100, 96])
@pytest.mark.parametrize("N,K", [(256, 128), (4096, 4096), (6144, 4096), (4096, 14336), (1000, 512)])
def test_stream_gemm_fp8_weights(M, N, K):
    """y = x @ dequant(W_fp8)^T: the kernel must match a plain fp32 matmul against the
    SAME quantised weights (so the tolerance covers accumulation order only)."""
    from kvmini.models.llama import quantize_fp8_rows
    from kvmini.ops import hip_ops

    torch.manual_seed(5)
    W = (torch.randn(N, K, device=_dev()) * 0.05).bfloat16()
    W[3] *= 7.0  # rows with different scales
    x = torch.randn(M, K, device=_dev()).bfloat16()
    q, scale = quantize_fp8_rows(W)
    Wdq = q.float() * scale[:, None]
    ref = x.float() @ Wdq.T
    tol = 1e-2 * ref.abs().max().item() + 1e-2
    pw = hip_ops.PackedLinear(W, fp8=True)
    assert pw.wp.dtype == torch.uint8 and pw.wp.numel() == pw.n_pad * K and pw.wscale is not None
    y = hip_ops.linear(x, pw)
    assert y.shape == (M, N)
    assert (y.float() - ref).abs().max().item() <= tol
    s = hip_ops.linear(x, pw, allow_slabs=True)
    got = s.slabs.sum(0) if isinstance(s, hip_ops.Partial) else s.float()
    assert (got - ref).abs().max().item() <= tol
    # and the quantisation error itself stays at the e4m3 level (3 mantissa bits)
    full = x.float() @ W.float().T
    assert (ref - full).abs().max().item() <= 0.08 * full.abs().max().item() + 0.05


@pytest.mark.parametrize("M", [16, 48])
def test_stream_gemm_fp8_swiglu(M):
    from kvmini.models.llama import interleave_gate_up, quantize_fp8_rows
    from kvmini.ops import hip_ops

    torch.manual_seed(6)
    I, K = 1024, 512
    wg = (torch.randn(I, K, device=_dev()) * 0.05).bfloat16()
    wu = (torch.randn(I, K, device=_dev()) * 0.05).bfloat16()
    x = torch.randn(M, K, device=_dev()).bfloat16()
    act = hip_ops.linear_swiglu(x, hip_ops.PackedLinear(interleave_gate_up(wg, wu).contiguous(), fp8=True))
    dq = lambda w: (lambda q, s: q.float() * s[:, None])(*quantize_fp8_rows(w))  # noqa: E731
    ref = torch.nn.functional.silu(x.float() @ dq(wg).T) * (x.float() @ dq(wu).T)
    assert act.shape == (M, I)
    assert (act.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item() + 1e-2


def test_engine_fp8_matches_fake_quantised_oracle():
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import fake_quantize_fp8, get_config, random_weights
    from kvmini.models.reference import dense_logits

    cfg = get_config("small-hd128")
    w = random_weights(cfg, "cuda:0", torch.bfloat16, seed=4)
    oracle = random_weights(cfg, "cuda:0", torch.bfloat16, seed=4)
    bytes_bf16 = oracle.nbytes()
    fake_quantize_fp8(oracle)
    proj_bf16 = sum(t.numel() * 2 for lw in oracle.layers for t in (lw.wqkv, lw.wo, lw.wgu, lw.wdown))
    # a 1024-token step budget: wide steps run on the tiled kernel, which reads the same packed
    # e4m3 bytes (widened in registers) — no dequantised or row-major copy is kept for them
    eng = LLMEngine(EngineConfig(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=1024, max_model_len=2048, num_blocks=256, kernels="hip", quantization="fp8"), weights=w)
    eng.warmup()
    assert eng.weights.quantization == "fp8" and eng.weights.layers[0].wqkv.wscale is not None
    proj_fp8 = sum(t.wp.numel() * t.wp.element_size() + t.wscale.numel() * 4 for lw in eng.weights.layers for t in (lw.wqkv, lw.wo, lw.wgu, lw.wdown))
    assert proj_fp8 <= 0.55 * proj_bf16, (proj_fp8, proj_bf16)
    assert not any(hasattr(t, "w_full") for lw in eng.weights.layers for t in (lw.wqkv, lw.wo, lw.wgu, lw.wdown))
    assert eng.weights.nbytes() < 0.8 * bytes_bf16  # embedding and LM head stay bf16
    prompts = [[1, 50, 60, 70, 80], list(range(1, 30)), [1] + [(7 * i) % 900 + 20 for i in range(299)]]
    outs = eng.generate(prompts, SamplingParams(max_tokens=8, ignore_eos=True))
    for p, o in zip(prompts, outs):
        ids = list(p)
        for tok in o:
            lg = dense_logits(oracle, ids)[-1]
            top = lg.max().item()
            assert lg[tok].item() >= top - 0.05 * abs(top) - 0.05, (p, o)
            ids.append(tok)


def test_speculative_decoding_on_gpu_matches_plain_greedy():
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    kw = dict(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, num_blocks=128, kernels="hip")
    prompts = [[1, 50, 60, 70, 80], list(range(1, 41)), [1, 9, 8]]
    params = SamplingParams(max_tokens=10, ignore_eos=True)
    base = LLMEngine(EngineConfig(**kw))
    base.warmup()
    want = base.generate(prompts, params)
    spec = LLMEngine(EngineConfig(**kw, draft_model="small-hd128-draft", num_speculative_tokens=2))
    spec.warmup()
    assert spec.spec.draft.graphs, "draft runner should replay HIP graphs"
    got = spec.generate(prompts, params)
    # same kernels, same weights; verification batches 3 positions per sequence, so a
    # near-tie in the logits may resolve differently: compare with a margin via the lengths
    # and require exact equality on the first tokens
    assert [len(o) for o in got] == [len(o) for o in want]
    assert all(g[:3] == w[:3] for g, w in zip(got, want)), (got, want)
    assert spec.stats.spec_proposed > 0


@pytest.mark.parametrize("budget", [256, 1024])
def test_engine_wide_prefill_paths_match_oracle(budget):
    """A 200-token prompt is prefilled in one chunk wider than the streaming GEMM's rows: the
    tiled MFMA kernel (on the same packed weights, no second copy) must agree with the dense oracle."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights
    from kvmini.models.reference import dense_logits

    cfg = get_config("small-hd128")
    w = random_weights(cfg, "cuda:0", torch.bfloat16, seed=7)
    oracle = random_weights(cfg, "cuda:0", torch.bfloat16, seed=7)
    eng = LLMEngine(EngineConfig(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=budget, max_model_len=512, num_blocks=256, kernels="hip"), weights=w)
    eng.warmup()
    assert not hasattr(eng.weights.layers[0].wqkv, "w_full")
    prompt = [1] + [(13 * i) % 900 + 20 for i in range(199)]
    out = eng.generate([prompt, [1, 2, 3]], SamplingParams(max_tokens=4, ignore_eos=True))[0]
    ids = list(prompt)
    for tok in out:
        lg = dense_logits(oracle, ids)[-1]
        top = lg.max().item()
        assert lg[tok].item() >= top - 0.05 * abs(top) - 0.05
        ids.append(tok)


@pytest.mark.parametrize("min_pos", ["0", "1000000"])
def test_engine_prompt_steps_on_the_eight_wave_attention_kernel(monkeypatch, min_pos):
    """The runner pads every chunk's tiles to groups of four and sends a step to the 8-wave attention
    kernel when its groups fill the CUs and its chunks start deep enough; with both thresholds
    lowered a small model takes that path for every multi-tile chunk (a 300-token prompt in 128-token
    chunks beside a short one and running decodes) and must still follow the dense oracle.  With the
    position threshold out of reach the tile list is still padded but the other kernels serve it
    (pairs stay aligned, unused tiles are skipped): the same answers."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights
    from kvmini.models.reference import dense_logits

    monkeypatch.setenv("KVMINI_QUADS_MIN_GROUPS", "1")
    monkeypatch.setenv("KVMINI_QUADS_MIN_POS", min_pos)
    cfg = get_config("small-hd128")
    w = random_weights(cfg, "cuda:0", torch.bfloat16, seed=7)
    oracle = random_weights(cfg, "cuda:0", torch.bfloat16, seed=7)
    eng = LLMEngine(EngineConfig(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=128, max_model_len=512, num_blocks=256, kernels="hip"), weights=w)
    eng.warmup()
    assert eng.runner.tile_quads and eng.runner.max_tiles % 4 == 0
    calls = {"quads": 0, "other": 0}
    real = eng.runner.ops.prefill_attention

    def counted(*a, **kw):
        calls["quads" if kw.get("quads") else "other"] += 1
        return real(*a, **kw)

    monkeypatch.setattr(eng.runner.ops, "prefill_attention", counted)
    prompts = [[1] + [(13 * i) % 900 + 20 for i in range(299)], [1, 2, 3], [1] + [(5 * i) % 700 + 30 for i in range(40)]]
    outs = eng.generate(prompts, SamplingParams(max_tokens=4, ignore_eos=True))
    assert (calls["quads"] > 0) == (min_pos == "0")  # (steps for the other kernels replay captured graphs: not counted here)
    for p, out in zip(prompts, outs):
        ids = list(p)
        for tok in out:
            lg = dense_logits(oracle, ids)[-1]
            top = lg.max().item()
            assert lg[tok].item() >= top - 0.05 * abs(top) - 0.05
            ids.append(tok)


def test_native_sampler_and_rccl_gather_on_gpu():
    import time

    import numpy as np

    from kvmini.telemetry import native_rt as rt

    assert rt.load() is not None, "libkvmini_rt.so must be built in-tree"
    assert rt.native_gpu_count() >= 1
    s = rt.NativeSampler(0.01, gpu_indices=[0]).start()
    x = torch.randn(4096, 4096, device="cuda:0")
    t0 = time.time()
    while time.time() - t0 < 0.3:
        (x @ x).sum().item()
    samples = s.stop()
    assert len(samples) >= 10, len(samples)
    dts = np.diff([p["ts_s"] for p in samples])
    assert np.median(dts) < 0.03  # the C++ thread keeps its cadence while Python is busy
    watts = [p["watts"] for p in samples if p["watts"] is not None]
    assert watts and 10 < max(watts) < 2000
    assert any(p["per_gpu"][0]["vram_used"] for p in samples)
    g = rt.NativeGather(0)  # world of one: the collective still runs through RCCL
    local = np.arange(32, dtype=np.float64).reshape(2, 16)
    table = g.gather(local)
    g.close()
    assert table.shape == (1, 2, 16) and np.array_equal(table[0], local)


def test_sampler_loaded_before_torch_keeps_one_hip_runtime():
    """The harness starts the C++ sampler before anything imports torch; HIP kernels of the
    extension must still launch afterwards (a sampler library that dragged the system HIP
    runtime in made them fail with 'no ROCm-capable device')."""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from kvmini.telemetry import native_rt as rt\n"
        "s = rt.NativeSampler(0.01, gpu_indices=[0]).start()\n"
        "import torch\n"
        "from kvmini.telemetry.energy import PowerSample, integrate_series_batched\n"
        "series = {0: [PowerSample(float(t), 360.0) for t in range(0, 11)]}\n"
        "out = integrate_series_batched(series, 0.0, 10.0, engine='hip')\n"
        "s.stop()\n"
        "maps = open('/proc/self/maps').read()\n"
        "n_hip = len({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l})\n"
        "print('WH', out[0], 'NHIP', n_hip)\n"
    ) % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert "WH 1.0" in r.stdout and "NHIP 1" in r.stdout, r.stdout


@pytest.mark.parametrize("rep", [1, 3, 4, 6, 8])
def test_fused_decode_attention_matches_two_kernel_path(rep):
    """RoPE + KV write + attention in one launch == rope_kv_write_slabs followed by
    paged_attention, for a decode-only step (one new token per sequence, padding rows included)."""
    from kvmini.models.llama import LlamaConfig, rope_tables
    from kvmini.ops import hip_ops

    torch.manual_seed(7)
    nkv, hd, bs, nb = 2, 128, 16, 96
    nh = nkv * rep
    n_seqs, max_blocks, T, S, h_in = 16, 6, 16, 5, 512
    width = (nh + 2 * nkv) * hd
    cos, sin = rope_tables(LlamaConfig(head_dim=hd), 128, _dev())
    ctx = torch.tensor([1, 2, 16, 17, 33, 64, 80, 0, 5, 48, 96, 31, 32, 0, 9, 65], device=_dev(), dtype=torch.int32)
    positions = (ctx - 1).clamp(min=0)
    tok_seq = torch.arange(T, device=_dev(), dtype=torch.int32)
    bt = torch.randperm(nb, device=_dev())[: n_seqs * max_blocks].view(n_seqs, max_blocks).int()
    slots = torch.where(ctx > 0, bt[torch.arange(T, device=_dev()), (positions // bs).long()] * bs + positions % bs, torch.full_like(ctx, -1)).int()
    slabs = torch.randn(S, T, width, device=_dev()) * 0.4
    ss = torch.rand(T, 1, device=_dev()) * h_in + 1.0

    def caches():
        g = torch.Generator(device=_dev()); g.manual_seed(1)
        return (torch.randn(nb, nkv, bs, hd, device=_dev(), generator=g).bfloat16(), torch.randn(nb, nkv, bs, hd, device=_dev(), generator=g).bfloat16())

    kc1, vc1 = caches()
    part = hip_ops.Partial(slabs, ss, 1e-5, h_in)
    q = hip_ops.rope_kv_write(part, positions, cos, sin, kc1, vc1, slots, nh, nkv, hd)
    ref = hip_ops.paged_attention(q, kc1, vc1, bt, tok_seq, ctx, 1 / math.sqrt(hd))
    kc2, vc2 = caches()
    assert hip_ops.supports_fused_decode_attention(part, nh, kc2)
    out = hip_ops.rope_attention_fused(part, positions, cos, sin, kc2, vc2, slots, bt, tok_seq, ctx, nh, 1 / math.sqrt(hd))
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)  # identical cache writes
    assert (out.float() - ref.float()).abs().max().item() <= 2e-2
    assert out[7].abs().max().item() == 0 and out[13].abs().max().item() == 0  # padding rows


@pytest.mark.parametrize("graphs,custom", [(False, True), (True, True), (False, False), (True, False)])
def test_engine_tp_collective_path_on_one_gpu(graphs, custom):
    """The tensor-parallel code path with a one-rank group must reproduce the plain engine's
    tokens: the fused peer-memory epilogue (slab sum -> all-reduce -> residual -> sums of squares;
    the rank is its own xGMI peer) and the RCCL all-reduce, eager and captured in HIP graphs."""
    import torch.distributed as dist

    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29793")
    own_pg = not dist.is_initialized()
    if own_pg:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        _tp_one_rank_body(graphs, custom)
    finally:
        if own_pg:
            dist.destroy_process_group()


def _tp_one_rank_body(graphs, custom):
    import torch.distributed as dist

    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights

    common = dict(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, num_blocks=128, kernels="hip")
    prompts = [[1, 50, 60, 70, 80], list(range(1, 30))]
    params = SamplingParams(max_tokens=8, ignore_eos=True)
    base = LLMEngine(EngineConfig(**common, use_graphs=False), weights=random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=3))
    want = base.generate(prompts, params)
    eng = LLMEngine(EngineConfig(**common, use_graphs=graphs, tp_graphs=graphs, tp_rccl_graphs=graphs, tp_custom_allreduce=custom, tp_force_collectives=True),
                    weights=random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=3), tp_group=dist.group.WORLD)
    assert eng.runner.tp_active and (eng.runner.xgmi is not None) == custom
    if graphs:
        eng.warmup()
        assert eng.runner.graphs, "graphs with the all-reduces inside were not captured"
    assert eng.generate(prompts, params) == want
    if custom:
        assert eng.runner.xgmi.calls > 0 and eng.runner.xgmi.calls % 2 == 0 and eng.runner.xgmi.errors() == 0
        assert eng.metrics()["tp_allreduce"] == "xgmi-peer-memory"


@pytest.mark.parametrize("rep,nkv,chunks", [(r, 2, [(0, 0, 40), (1, 37, 23), (2, 0, 5), (3, 300, 33)]) for r in (1, 2, 3, 4, 6, 7, 8)]
                         + [(r, 8, [(0, 0, 600), (1, 37, 250), (2, 0, 5), (3, 300, 200)]) for r in (1, 2, 3, 4, 8)])
def test_prefill_attention_tiles_match_reference(rep, nkv, chunks):
    """The tiled MFMA prompt-attention kernel against the per-token reference: chunks that
    start at position 0, mid-block continuations of an earlier chunk, short chunks, and rows
    that belong to no tile (left untouched).  The larger case (8 kv heads, 68 tiles) reaches
    the 32-queries-per-workgroup path (32x32x16 MFMA; head groups of 1-4, so also waves that own
    no head), with pairs that straddle chunk boundaries, a partial second tile and an odd tile
    count."""
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(11)
    hd, bs, nb = 128, 16, 200
    nh = nkv * rep
    n_seqs, max_blocks = 4, 40
    kc, vc = _rand_cache(nb, nkv, bs, hd), _rand_cache(nb, nkv, bs, hd)
    bt = torch.randperm(nb, device=_dev())[: n_seqs * max_blocks].view(n_seqs, max_blocks).int()
    # chunks: (block-table row, first position, tokens): flat rows in this order, one decode row between
    rows, tok_seq, ctx, tiles = 0, [], [], []
    for k, (seq, p0, n) in enumerate(chunks):
        if k == 2:  # a row owned by nobody: must keep its sentinel value
            tok_seq.append(0); ctx.append(0); rows += 1
        for off in range(0, n, 16):
            tiles.append((rows + off, min(16, n - off), p0 + off, seq))
        tok_seq += [seq] * n
        ctx += [p0 + j + 1 for j in range(n)]
        rows += n
    tiles += [(0, 0, 0, 0)] * 3  # unused tiles
    T = rows
    q = torch.randn(T, nh, hd, device=_dev()).bfloat16()
    tok_seq_t = torch.tensor(tok_seq, device=_dev(), dtype=torch.int32)
    ctx_t = torch.tensor(ctx, device=_dev(), dtype=torch.int32)
    scale = 1 / math.sqrt(hd)
    ref = torch_ops.paged_attention(q, kc, vc, bt, tok_seq_t, ctx_t, scale)
    out = torch.full((T, nh * hd), 7.0, device=_dev(), dtype=torch.bfloat16)
    tiles_t = torch.tensor(tiles, device=_dev(), dtype=torch.int32).T.contiguous()
    assert hip_ops.supports_prefill_attention(nh, kc)
    hip_ops.prefill_attention(q, kc, vc, bt, tiles_t, out, scale)
    owned = ctx_t > 0
    assert (out[owned].float() - ref[owned].float()).abs().max().item() <= 2.5e-2
    assert (out[~owned] == 7.0).all()


@pytest.mark.parametrize("seed,rep,kv8,quads", [(1, 4, False, False), (2, 4, True, False), (3, 1, False, False), (4, 2, False, False), (5, 8, False, False), (6, 6, True, False),
                                               (7, 3, True, False), (8, 4, False, True), (9, 4, True, True), (10, 1, False, True), (11, 3, True, True), (12, 2, False, True)])
def test_prefill_attention_random_chunk_layouts(seed, rep, kv8, quads):
    """Random chunk layouts (starts anywhere, lengths 1..300, continuation chunks of earlier ones,
    enough tiles that head groups of at most 4 take the 32-query path) against the per-token
    reference: partial tiles, pairs that are and are not neighbours, passes that end inside a
    cache block, first steps that already reach the diagonal."""
    from kvmini.ops import hip_ops, torch_ops

    rng = random.Random(seed)
    torch.manual_seed(seed)
    nkv, hd, bs = 8, 128, 16
    nh = nkv * rep
    n_seqs, max_blocks = 6, 40
    nb = n_seqs * max_blocks + 8
    bt = torch.randperm(nb, device=_dev())[: n_seqs * max_blocks].view(n_seqs, max_blocks).int()
    sc = None
    if kv8:
        kc, vc = _fp8_cache(nb, nkv, bs, hd, 1), _fp8_cache(nb, nkv, bs, hd, 2)
        sc = (torch.rand(2, nkv, device=_dev()) + 0.5).contiguous()
    else:
        kc, vc = _rand_cache(nb, nkv, bs, hd), _rand_cache(nb, nkv, bs, hd)
    rows, tok_seq, ctx, tiles = 0, [], [], []
    pos = [0] * n_seqs  # a sequence's chunks continue where its last one ended (or skip ahead: a prefix-cache hit)
    while len(tiles) < 70:
        seq = rng.randrange(n_seqs)
        p0 = pos[seq] + rng.choice((0, 0, 0, rng.randrange(0, 50)))
        n = min(rng.choice((rng.randint(1, 15), rng.randint(16, 48), rng.randint(49, 300))), max_blocks * bs - p0)
        if n <= 0:
            continue
        for off in range(0, n, 16):
            tiles.append((rows + off, min(16, n - off), p0 + off, seq))
        tok_seq += [seq] * n
        ctx += [p0 + j + 1 for j in range(n)]
        rows += n
        pos[seq] = p0 + n
    q = torch.randn(rows, nh, hd, device=_dev()).bfloat16()
    tok_seq_t = torch.tensor(tok_seq, device=_dev(), dtype=torch.int32)
    ctx_t = torch.tensor(ctx, device=_dev(), dtype=torch.int32)
    scale = 1 / math.sqrt(hd)
    kw = {"kv_scale": sc} if kv8 else {}
    ref = torch_ops.paged_attention(q, kc, vc, bt, tok_seq_t, ctx_t, scale, **kw)
    out = torch.zeros((rows, nh * hd), device=_dev(), dtype=torch.bfloat16)
    if quads:  # the 8-wave kernel: groups of four neighbouring tiles, chunks padded to a multiple of four
        tiles = hip_ops.pad_tiles_to_quads(tiles)
        assert len(tiles) % 4 == 0 and all(t[1] == 0 or t[3] == tiles[4 * (i // 4)][3] for i, t in enumerate(tiles))
    hip_ops.prefill_attention(q, kc, vc, bt, torch.tensor(tiles, device=_dev(), dtype=torch.int32).T.contiguous(), out, scale, quads=quads, **kw)
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref.float()).abs().max().item() <= (4e-2 if kv8 else 2.5e-2)


@pytest.mark.parametrize("seed,rep,kv8,n_split", [(1, 4, False, 2), (2, 4, True, 8), (3, 1, False, 4), (4, 2, True, 32), (5, 8, False, 8), (6, 6, True, 4), (7, 3, False, 16),
                                                 (8, 7, False, 2), (9, 4, False, 32)])
def test_prefill_attention_split_random_chunk_layouts(seed, rep, kv8, n_split):
    """The split form of the tile kernel (every tile's key range in n_split shares + a combine launch) against the
    per-token reference: short chunks at random depths of contexts of up to 3200 tokens — shares that are empty for
    the shallow chunks, shares that end on the diagonal, partial tiles, pairs that are and are not neighbours,
    unused tiles inside and behind the list."""
    from kvmini.ops import hip_ops, torch_ops

    rng = random.Random(seed)
    torch.manual_seed(seed)
    nkv, hd, bs = 8, 128, 16
    nh = nkv * rep
    n_seqs, max_blocks = 5, 200
    nb = n_seqs * max_blocks + 8
    bt = torch.randperm(nb, device=_dev())[: n_seqs * max_blocks].view(n_seqs, max_blocks).int()
    sc = None
    if kv8:
        kc, vc = _fp8_cache(nb, nkv, bs, hd, 1), _fp8_cache(nb, nkv, bs, hd, 2)
        sc = (torch.rand(2, nkv, device=_dev()) + 0.5).contiguous()
    else:
        kc, vc = _rand_cache(nb, nkv, bs, hd), _rand_cache(nb, nkv, bs, hd)
    rows, tok_seq, ctx, tiles = 0, [], [], []
    for seq in rng.sample(range(n_seqs), 4):
        n = rng.choice((rng.randint(1, 15), rng.randint(17, 40), rng.randint(41, 90)))
        p0 = rng.choice((0, rng.randrange(0, 64), rng.randrange(500, max_blocks * bs - n), max_blocks * bs - n))
        if rng.random() < 0.5:
            tiles.append((0, 0, 0, 0))  # an unused tile inside the list
        for off in range(0, n, 16):
            tiles.append((rows + off, min(16, n - off), p0 + off, seq))
        tok_seq += [seq] * n
        ctx += [p0 + j + 1 for j in range(n)]
        rows += n
    n_tiles = len(tiles)
    tiles += [(0, 0, 0, 0)] * 5  # the engine's tile arrays are wider than a step's list
    q = torch.randn(rows, nh, hd, device=_dev()).bfloat16()
    scale = 1 / math.sqrt(hd)
    kw = {"kv_scale": sc} if kv8 else {}
    ref = torch_ops.paged_attention(q, kc, vc, bt, torch.tensor(tok_seq, device=_dev(), dtype=torch.int32), torch.tensor(ctx, device=_dev(), dtype=torch.int32), scale, **kw)
    out = torch.zeros((rows, nh * hd), device=_dev(), dtype=torch.bfloat16)
    hip_ops.prefill_attention_split(q, kc, vc, bt, torch.tensor(tiles, device=_dev(), dtype=torch.int32).T.contiguous(), n_tiles, out, scale, n_split, **kw)
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref.float()).abs().max().item() <= (4e-2 if kv8 else 2.5e-2)
    # and the unsplit kernel agrees with it closely (same pass, other key ranges)
    one = torch.zeros_like(out)
    hip_ops.prefill_attention(q, kc, vc, bt, torch.tensor(tiles[:n_tiles], device=_dev(), dtype=torch.int32).T.contiguous(), one, scale, **kw)
    assert (out.float() - one.float()).abs().max().item() <= 2e-2


def test_prefill_split_plan():
    from kvmini.ops import hip_ops

    assert hip_ops.prefill_split_plan(16, 32, 8, 131072) == 8   # 256 new tokens: 64 groups x 8 shares = two workgroups per CU
    assert hip_ops.prefill_split_plan(4, 32, 8, 131072) == 32   # 64 new tokens
    assert hip_ops.prefill_split_plan(4, 32, 8, 4096) == 4      # shares of at least 1024 keys
    assert hip_ops.prefill_split_plan(4, 32, 8, 1500) == 0
    assert hip_ops.prefill_split_plan(128, 32, 8, 131072) == 0  # a long chunk fills the chip by itself
    assert hip_ops.prefill_split_plan(8, 64, 8, 65536) == 8     # head groups of 8: one tile per group


@pytest.mark.parametrize("quads", [False, True])
@pytest.mark.parametrize("kv8", [False, True])
def test_prefill_attention_follows_a_maximum_that_jumps_mid_pass(kv8, quads):
    """The 32-query path exponentiates against a reference that may lag the running maximum and
    re-references only when a query's scores exceed it by more than 2^6.  On bounded random data
    that branch never fires after the first step, so this case makes it fire: the keys of later
    blocks are several times larger (scores jump by tens of log2 units at blocks 8, 20 and 33 of a
    700-token prompt), for a chunk from position 0 and for a continuation chunk whose pass starts
    below the jumps."""
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(5)
    rep, nkv, hd, bs, nb = 4, 8, 128, 16, 120
    nh = nkv * rep
    n_seqs, max_blocks = 2, 48
    bt = torch.randperm(nb, device=_dev())[: n_seqs * max_blocks].view(n_seqs, max_blocks).int()
    kf = torch.randn(nb, nkv, bs, hd, device=_dev()) * 0.5
    vf = torch.randn(nb, nkv, bs, hd, device=_dev()) * 0.5
    for seq in range(n_seqs):
        for j, gain in ((8, 3.0), (20, 2.0), (33, 2.0)):  # cumulative: x3, x6, x12 from those blocks on
            kf[bt[seq, j:].long()] *= gain
    sc = None
    if kv8:
        sc = torch.stack([kf.abs().amax(dim=(0, 2, 3)) / 400.0, vf.abs().amax(dim=(0, 2, 3)) / 400.0]).float().contiguous()
        kc = (kf / sc[0].view(1, nkv, 1, 1)).to(torch.float8_e4m3fn)
        vc = (vf / sc[1].view(1, nkv, 1, 1)).to(torch.float8_e4m3fn)
    else:
        kc, vc = kf.bfloat16(), vf.bfloat16()
    chunks = [(0, 0, 700), (1, 100, 600)]
    rows, tok_seq, ctx, tiles = 0, [], [], []
    for seq, p0, n in chunks:
        for off in range(0, n, 16):
            tiles.append((rows + off, min(16, n - off), p0 + off, seq))
        tok_seq += [seq] * n
        ctx += [p0 + j + 1 for j in range(n)]
        rows += n
    q = torch.randn(rows, nh, hd, device=_dev()).bfloat16()
    tok_seq_t = torch.tensor(tok_seq, device=_dev(), dtype=torch.int32)
    ctx_t = torch.tensor(ctx, device=_dev(), dtype=torch.int32)
    scale = 1 / math.sqrt(hd)
    kw = {"kv_scale": sc} if kv8 else {}
    ref = torch_ops.paged_attention(q, kc, vc, bt, tok_seq_t, ctx_t, scale, **kw)
    out = torch.zeros((rows, nh * hd), device=_dev(), dtype=torch.bfloat16)
    assert nkv * len(tiles) >= 512  # the 32-query path
    if quads:
        tiles = hip_ops.pad_tiles_to_quads(tiles)
    hip_ops.prefill_attention(q, kc, vc, bt, torch.tensor(tiles, device=_dev(), dtype=torch.int32).T.contiguous(), out, scale, quads=quads, **kw)
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref.float()).abs().max().item() <= (4e-2 if kv8 else 2.5e-2)


def test_engine_pads_unsupported_head_widths_to_128():
    """A model with 80-dim heads runs on the 128-dim attention kernels (zero-padded heads)
    and still follows the dense oracle built from the un-padded weights."""
    from dataclasses import replace

    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights
    from kvmini.models.reference import dense_logits

    cfg = replace(get_config("small-hd128"), name="small-hd80", head_dim=80)
    w = random_weights(cfg, "cuda:0", torch.bfloat16, seed=5)
    oracle = random_weights(cfg, "cuda:0", torch.bfloat16, seed=5)
    eng = LLMEngine(EngineConfig(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, num_blocks=128, kernels="hip"), weights=w)
    assert eng.mcfg.head_dim == 128 and eng.mcfg.rope_dim == 80
    eng.warmup()
    prompt = [1] + [(11 * i) % 900 + 20 for i in range(40)]
    out = eng.generate([prompt, [1, 2, 3]], SamplingParams(max_tokens=6, ignore_eos=True))[0]
    ids = list(prompt)
    for tok in out:
        lg = dense_logits(oracle, ids)[-1]
        top = lg.max().item()
        assert lg[tok].item() >= top - 0.05 * abs(top) - 0.05
        ids.append(tok)


def test_engine_replicates_kv_heads_for_unsupported_grouping():
    """5 query heads per kv head is outside the attention kernels' set: the runner stores every
    K/V head 5 times (ratio 1) and the engine still follows the oracle of the original model."""
    from dataclasses import replace

    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights
    from kvmini.models.reference import dense_logits

    cfg = replace(get_config("small-hd128"), name="small-gqa5", num_heads=10, num_kv_heads=2)
    w = random_weights(cfg, "cuda:0", torch.bfloat16, seed=6)
    oracle = random_weights(cfg, "cuda:0", torch.bfloat16, seed=6)
    eng = LLMEngine(EngineConfig(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, num_blocks=128, kernels="hip"), weights=w)
    assert eng.mcfg.num_kv_heads == 10
    eng.warmup()
    prompt = [1] + [(7 * i) % 900 + 20 for i in range(40)]
    out = eng.generate([prompt, [1, 2, 3]], SamplingParams(max_tokens=6, ignore_eos=True))[0]
    ids = list(prompt)
    for tok in out:
        lg = dense_logits(oracle, ids)[-1]
        top = lg.max().item()
        assert lg[tok].item() >= top - 0.05 * abs(top) - 0.05
        ids.append(tok)


@pytest.mark.gpu
def test_embed_fed_matches_reference_and_pipelined_engine_matches_sync():
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(0)
    V, H, T = 1000, 512, 48
    table = torch.randn(V, H, device="cuda", dtype=torch.bfloat16)
    tokens = torch.randint(0, V, (T,), device="cuda", dtype=torch.int32)
    prev = torch.randint(0, V, (64,), device="cuda", dtype=torch.int64)
    feed = torch.zeros(T, device="cuda", dtype=torch.int32)
    feed[::3] = torch.randint(1, 65, (len(feed[::3]),), device="cuda", dtype=torch.int32)
    feed[5] = 65 + 7  # beyond prev: falls back to the staged token
    tokens[7] = V + 3  # outside the vocabulary: row 0
    got = hip_ops.embed_tokens(table, tokens, feed, prev)
    want = torch_ops.embed_tokens(table, tokens, feed, prev)
    assert torch.equal(got, want)
    assert torch.equal(got[3], table[prev[feed[3] - 1]]) and torch.equal(got[7], table[0]) and torch.equal(got[5], table[tokens[5]])

    # engine level: launch-ahead stepping (graphs, device-side token feed) = one step at a time
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    prompts = [[1] + list(range(100 + 3 * i, 110 + 4 * i)) for i in range(6)] + [[1] + list(range(300, 420))]
    outs = {}
    for mode in (False, True):
        eng = LLMEngine(EngineConfig(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=512, kernels="hip", async_scheduling=mode, num_blocks=256))
        eng.warmup()
        seqs = [eng.make_sequence(p, SamplingParams(max_tokens=24 + 3 * i, ignore_eos=True)) for i, p in enumerate(prompts)]
        for s in seqs[:3]:
            eng.scheduler.add(s)
        step = 0
        while eng.scheduler.has_work() or step < 8:
            if step == 4:
                for s in seqs[3:]:
                    eng.scheduler.add(s)
            eng.step()
            step += 1
        outs[mode] = [s.output_ids for s in seqs]
        assert (eng.pipelined_steps > 20) == mode
        assert all(len(s.output_ids) == 24 + 3 * i for i, s in enumerate(seqs))
    assert outs[True] == outs[False]


@pytest.mark.gpu
@pytest.mark.parametrize("nw", [4, 7, 8])
@pytest.mark.parametrize("M", [16, 64])
def test_stream_gemm3_waves_per_workgroup(C, nw, M):
    torch.manual_seed(nw * 10 + M)
    N, K = 16 * nw * 6, 1024
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    wp = C.pack_weight(w)
    ref = x.float() @ w.float().T
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    C.stream_gemm3(wp, N, K, x, y, 0, 2, 1, K // 256, None, 0.0, None, nw)
    assert torch.allclose(y.float(), ref, atol=0.05, rtol=0.02)
    splits, gps = C.stream_gemm3_plan(N, K, 2, 8, 64, nw)
    slabs = torch.empty(splits, M, N, device="cuda", dtype=torch.float32)
    C.stream_gemm3(wp, N, K, x, slabs, 1, 2, splits, gps, None, 0.0, None, nw)
    assert splits > 1 and torch.allclose(slabs.sum(0), ref, atol=0.02, rtol=0.01)


@pytest.mark.gpu
def test_scale_rows_matches_reference(C):
    torch.manual_seed(3)
    T, H = 37, 4096
    x = torch.randn(T, H, device="cuda").bfloat16()
    ss = (x.float() ** 2).view(T, 8, 512).sum(-1).contiguous()
    y = torch.empty_like(x)
    C.scale_rows(x, ss, 1e-5, y)
    ref = (x.float() * torch.rsqrt((x.float() ** 2).mean(-1, keepdim=True) + 1e-5)).bfloat16()
    assert torch.allclose(y.float(), ref.float(), atol=1e-2, rtol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [96, 128, 256])
@pytest.mark.parametrize("nw", [4, 7, 8])
def test_stream_gemm3_wide_rows(C, M, nw):
    """96 / 128 / 256 token rows in one pass over the weights (6 / 8 / 16 row tiles per wave)."""
    from kvmini.models.llama import interleave_gate_up
    from kvmini.ops import torch_ops

    torch.manual_seed(M + nw)
    N, K = 16 * nw * 4, 1024
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    wp = C.pack_weight(w)
    ref = x.float() @ w.float().T
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    C.stream_gemm3(wp, N, K, x, y, 0, 2, 1, K // 256, None, 0.0, None, nw)
    assert torch.allclose(y.float(), ref, atol=0.06, rtol=0.02)
    splits, gps = C.stream_gemm3_plan(N, K, 2, 4, 64, nw)
    slabs = torch.empty(splits, M, N, device="cuda", dtype=torch.float32)
    C.stream_gemm3(wp, N, K, x, slabs, 1, 2, splits, gps, None, 0.0, None, nw)
    assert splits > 1 and torch.allclose(slabs.sum(0), ref, atol=0.02, rtol=0.01)
    # SwiGLU epilogue + folded row scale
    wg, wu = w[: N // 2], w[N // 2 :]
    wgu = interleave_gate_up(wg, wu).contiguous()
    ss = (x.float() ** 2).view(M, K // 512, 512).sum(-1).contiguous()
    act = torch.empty(M, N // 2, device="cuda", dtype=torch.bfloat16)
    C.stream_gemm3(C.pack_weight(wgu), N, K, x, act, 2, 2, 1, K // 256, ss, 1e-5, None, nw)
    xn = x.float() * torch.rsqrt((x.float() ** 2).mean(-1, keepdim=True) + 1e-5)
    ref_act = torch.nn.functional.silu(xn @ wg.float().T) * (xn @ wu.float().T)
    assert torch.allclose(act.float(), ref_act, atol=0.08, rtol=0.05)


@pytest.mark.gpu
@pytest.mark.parametrize("temp,top_k,top_p", [(0.7, 0, 1.0), (1.0, 50, 1.0), (1.3, 0, 0.9), (0.9, 40, 0.8), (1.0, 1, 1.0)])
def test_sample_rows_kernel_draws_from_the_reference_distribution(C, temp, top_k, top_p):
    """Inverse-CDF check: with the uniform variates on a regular grid the kernel must hit every
    token of the nucleus N * p_i times (+-2), and nothing outside it.  Reference: top-k by value
    threshold, then top-p on the renormalised survivors (ties with the threshold value kept)."""
    torch.manual_seed(int(temp * 10) + top_k)
    V, N = 4000, 4000
    base = (torch.randn(V, device="cuda") * 3).bfloat16()
    x = base.float() / temp
    keep = torch.ones(V, dtype=torch.bool, device="cuda")
    if top_k:
        keep &= x >= torch.topk(x, top_k).values[-1]
    probs = torch.softmax(torch.where(keep, x, torch.full_like(x, float("-inf"))), -1)
    if top_p < 1.0:
        sp, si = torch.sort(probs, descending=True)
        n_keep = int(((torch.cumsum(sp, 0) - sp) < top_p).sum())
        keep &= probs >= sp[n_keep - 1]
        probs = torch.where(keep, probs, torch.zeros_like(probs))
        probs = probs / probs.sum()
    logits = base[None, :].repeat(N, 1).contiguous()
    rows = torch.arange(N, device="cuda")
    u = ((torch.arange(N, device="cuda", dtype=torch.float32) + 0.5) / N).contiguous()
    out = torch.full((N,), -1, dtype=torch.int64, device="cuda")
    C.sample_rows(logits, rows, torch.full((N,), temp, device="cuda"), torch.full((N,), top_k, dtype=torch.int32, device="cuda"), torch.full((N,), top_p, device="cuda"), u, out)
    assert int(out.min()) >= 0 and bool(keep[out].all())
    counts = torch.bincount(out, minlength=V).float()
    assert float((counts - probs * N).abs().max()) <= 2.5


@pytest.mark.gpu
def test_sample_rows_kernel_full_vocabulary_and_engine_path():
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    eng = LLMEngine(EngineConfig(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, kernels="hip", num_blocks=128))
    eng.warmup()
    greedy = eng.generate([[1, 5, 9, 11]], SamplingParams(max_tokens=12, ignore_eos=True))[0]
    steps0 = eng.pipelined_steps
    a = eng.generate([[1, 5, 9, 11]], SamplingParams(max_tokens=12, temperature=0.8, top_k=1, ignore_eos=True))[0]  # top_k = 1: deterministic
    b = eng.generate([[1, 5, 9, 11], [1, 7, 7]], SamplingParams(max_tokens=12, temperature=1.1, top_p=0.9, ignore_eos=True))
    assert a == greedy and eng.pipelined_steps > steps0
    assert all(len(o) == 12 and all(0 <= t < eng.mcfg.vocab_size for t in o) for o in b)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(4096, 11008), (6144, 4096), (2560, 10240)])
def test_split_k_linear_with_uneven_splits_matches_reference(n, k):
    """hip_ops.linear with split-K slabs on shapes whose K does not divide evenly into the
    planned splits (11008 = 43 groups of 256): the last split is shorter, the sum is exact."""
    from kvmini.ops import hip_ops

    torch.manual_seed(n + k)
    w = (torch.randn(n, k, device="cuda") * 0.02).bfloat16()
    x = torch.randn(32, k, device="cuda").bfloat16()
    nw, splits, gps = hip_ops._plan_slabs(n, k)
    assert splits * gps >= k // 256 > (splits - 1) * gps
    y = hip_ops.linear(x, hip_ops.PackedLinear(w), allow_slabs=True)
    assert isinstance(y, hip_ops.Partial) and y.slabs.shape[0] == splits
    ref = x.float() @ w.float().T
    assert torch.allclose(y.reduce().float(), ref, atol=0.06, rtol=0.02)


@pytest.mark.parametrize("M", [129, 254, 512, 1024, 2048])
def test_tiled_gemm_through_the_op_layer_matches_fp32(M):
    """More than 128 token rows go to the tiled MFMA kernel (csrc/wide_gemm.hip) on the packed
    weights — bf16 output with the folded-norm row scale, split-K slabs for the narrow
    projections, SwiGLU on the interleaved gate/up rows — against an fp32 matmul."""
    import torch.nn.functional as F

    from kvmini.ops import hip_ops as ops

    dev = _dev()
    torch.manual_seed(M)
    K, N = 1024, 2048
    x = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    pw = ops.PackedLinear(W)
    ref = x.float() @ W.float().T
    # plain
    y = ops.linear(x, pw)
    assert y.shape == (M, N) and (y.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
    # folded norm: row scale in the epilogue, or carried by the slabs
    ss = (x.float() ** 2).sum(-1, keepdim=True).contiguous()
    rs = torch.rsqrt(ss / K + 1e-5)
    y = ops.linear(ops.Normed(x, ss, 1e-5), pw)
    assert (y.float() - ref * rs).abs().max().item() <= 1e-2 * (ref * rs).abs().max().item()
    part = ops.linear(ops.Normed(x, ss, 1e-5), pw, allow_slabs=True)
    got = part.reduce().float() if isinstance(part, ops.Partial) else part.float()
    assert (got - ref * rs).abs().max().item() <= 1e-2 * (ref * rs).abs().max().item()
    # SwiGLU
    act = ops.linear_swiglu(x, pw)
    g = ref.view(M, N // 16, 2, 8)
    want = (F.silu(g[:, :, 0]) * g[:, :, 1]).reshape(M, N // 2)
    assert act.shape == (M, N // 2) and (act.float() - want).abs().max().item() <= 1.5e-2 * want.abs().max().item() + 1e-3
    # FP8 weights: same kernel, bytes widened in registers, per-feature scale in the epilogue
    from kvmini.models.llama import quantize_fp8_rows

    q, scale = quantize_fp8_rows(W)
    wq = q.float() * scale[:, None]
    ref8 = x.float() @ wq.T
    pw8 = ops.PackedLinear(W, fp8=True)
    # wide steps quantise the activations per token as well (FP8 x FP8 on the scaled MFMA): exact against the
    # product of the quantised operands, within the e4m3 rounding noise of the unquantised activations
    y8 = ops.linear(x, pw8)
    _x8, a_sc = ops.native().quantize_rows_fp8(x)
    xq = (x.float() / a_sc[:, None]).to(torch.float8_e4m3fn).float() * a_sc[:, None]
    ref88 = xq @ wq.T
    assert (y8.float() - ref88).abs().max().item() <= 1e-2 * ref88.abs().max().item()
    assert ((y8.float() - ref8).norm() / ref8.norm()).item() < 0.05
    # KVMINI_FP8_ACT=0: bf16 activations, the weight bytes widened in registers
    prev, ops.FP8_ACTIVATIONS = ops.FP8_ACTIVATIONS, False
    try:
        y8 = ops.linear(x, pw8)
    finally:
        ops.FP8_ACTIVATIONS = prev
    assert (y8.float() - ref8).abs().max().item() <= 1e-2 * ref8.abs().max().item()


@pytest.mark.parametrize("tile,variant", [((256, 256), 0), ((256, 256), 1), ((256, 256), 3), ((256, 128), 0), ((256, 128), 1), ((256, 128), 2), ((128, 256), 0), ((128, 256), 2),
                                          ((256, 64), 0), ((256, 64), 2), ((128, 128), 0), ((128, 128), 2)])
def test_tiled_gemm_every_tile_and_pipeline_shape(C, tile, variant):
    """Every compiled (workgroup tile, ring depth) of the tiled kernel: M not a multiple of the
    tile, three uneven K splits, token-tile-fastest and feature-tile-fastest block orders."""
    dev = _dev()
    torch.manual_seed(3)
    bm, bn = tile
    M, N, K = 300, 512, 448  # 7 x 64: odd stage counts exercise the pipeline's tail
    x = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    wp = C.pack_weight(W)
    ref = x.float() @ W.float().T
    nst = K // (32 * C.wide_gemm_stage_ksteps(bm, bn, variant))
    for mf in (True, False):
        y = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        C.wide_gemm(wp, N, K, x, y, 0, bm, bn, variant, 1, nst, None, 0.0, None, mf)
        assert (y.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
    sps = (nst + 2) // 3
    nsp = -(-nst // sps)
    sl = torch.full((nsp, M, N), float("nan"), device=dev, dtype=torch.float32)
    C.wide_gemm(wp, N, K, x, sl, 1, bm, bn, variant, nsp, sps, None, 0.0, None, True)
    assert (sl.sum(0) - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("bn,variant", [(224, 0), (224, 1), (224, 3), (192, 0), (192, 1), (192, 3)])
@pytest.mark.parametrize("fp8", [False, True])
def test_tiled_gemm_224_and_192_feature_tiles(C, bn, variant, fp8):
    """Feature tiles of 224 and 192 (whole rounds over the CUs for 28672 and 6144 features): the
    W pieces of a stage do not divide by the eight waves, surplus DMA slots repeat a piece.  bf16,
    slabs and SwiGLU epilogues against fp32, bf16 and FP8 weights."""
    import torch.nn.functional as F

    from kvmini.models.llama import quantize_fp8_rows

    if fp8 and variant != 0:
        pytest.skip("FP8 weights use 64-wide stages only")
    dev = _dev()
    torch.manual_seed(bn + variant)
    M, N, K = 300, 2 * bn, 448
    x = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    if fp8:
        q, scale = quantize_fp8_rows(W)
        wp, ws, wref = C.pack_weight_fp8(q.view(torch.uint8).contiguous()), scale.contiguous(), q.float() * scale[:, None]
    else:
        wp, ws, wref = C.pack_weight(W), None, W.float()
    ref = x.float() @ wref.T
    nst = K // (32 * C.wide_gemm_stage_ksteps(256, bn, variant))
    for mf in (True, False):
        y = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        C.wide_gemm(wp, N, K, x, y, 0, 256, bn, variant, 1, nst, None, 0.0, ws, mf)
        assert (y.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
    sps = (nst + 2) // 3
    nsp = -(-nst // sps)
    sl = torch.full((nsp, M, N), float("nan"), device=dev, dtype=torch.float32)
    C.wide_gemm(wp, N, K, x, sl, 1, 256, bn, variant, nsp, sps, None, 0.0, ws, True)
    assert (sl.sum(0) - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    act = torch.full((M, N // 2), float("nan"), device=dev, dtype=torch.bfloat16)
    C.wide_gemm(wp, N, K, x, act, 2, 256, bn, variant, 1, nst, None, 0.0, ws, True)
    g = ref.view(M, N // 16, 2, 8)
    want = (F.silu(g[:, :, 0]) * g[:, :, 1]).reshape(M, N // 2)
    assert (act.float() - want).abs().max().item() <= 1.5e-2 * want.abs().max().item() + 1e-3


@pytest.mark.parametrize("tile,variant", [((256, 128), 4), ((256, 128), 5), ((128, 128), 4), ((128, 128), 5), ((128, 256), 4)])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("K", [128, 448, 1024])
def test_tiled_gemm_role_split_rings(C, tile, variant, fp8, K):
    """The role-split form of the tiled kernel (half the waves copy W into a deep ring, half copy
    x into a shallow one): bf16, slab and SwiGLU epilogues against fp32 with bf16 and FP8 weights,
    M not a multiple of the tile, both block orders, K shorter than the W ring (128 = 2 stages),
    odd stage counts (448) and uneven K splits."""
    import torch.nn.functional as F

    from kvmini.models.llama import quantize_fp8_rows

    dev = _dev()
    torch.manual_seed(K + variant)
    bm, bn = tile
    M, N = 300, 2 * bn
    x = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    if fp8:
        q, scale = quantize_fp8_rows(W)
        wp, ws, wref = C.pack_weight_fp8(q.view(torch.uint8).contiguous()), scale.contiguous(), q.float() * scale[:, None]
    else:
        wp, ws, wref = C.pack_weight(W), None, W.float()
    ref = x.float() @ wref.T
    nst = K // (32 * C.wide_gemm_stage_ksteps(bm, bn, variant))
    assert nst == K // 64
    for mf in (True, False):
        y = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        C.wide_gemm(wp, N, K, x, y, 0, bm, bn, variant, 1, nst, None, 0.0, ws, mf)
        assert (y.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
    sps = (nst + 2) // 3
    nsp = -(-nst // sps)
    sl = torch.full((nsp, M, N), float("nan"), device=dev, dtype=torch.float32)
    C.wide_gemm(wp, N, K, x, sl, 1, bm, bn, variant, nsp, sps, None, 0.0, ws, True)
    assert (sl.sum(0) - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    ss = (x.float() ** 2).sum(-1, keepdim=True).contiguous()
    act = torch.full((M, N // 2), float("nan"), device=dev, dtype=torch.bfloat16)
    C.wide_gemm(wp, N, K, x, act, 2, bm, bn, variant, 1, nst, ss, 1e-5, ws, True)
    g = (ref * torch.rsqrt(ss / K + 1e-5)).view(M, N // 16, 2, 8)
    want = (F.silu(g[:, :, 0]) * g[:, :, 1]).reshape(M, N // 2)
    assert (act.float() - want).abs().max().item() <= 1.5e-2 * want.abs().max().item() + 1e-3


def test_xgmi_collectives_single_rank_is_its_own_peer(C):
    """The peer-memory collectives with a group of one rank (the rank's buffer is its own peer):
    all-gather, one-shot and two-shot all-reduce are identities, and the fused TP epilogue equals
    the single-GPU residual_reduce kernel bit for bit — the same kernels that W ranks run."""
    from kvmini.parallel.xgmi import XgmiComm

    dev = _dev()
    comm = XgmiComm(device=dev, world=1, rank=0)
    torch.manual_seed(0)
    tab = torch.randn(24, 16, device=dev, dtype=torch.float64)
    assert torch.equal(comm.all_gather(tab)[0], tab)
    for n, two in ((8 * 1237, False), (8 * 40000, True), (8 * 40000, None)):
        x = torch.randn(n, device=dev).bfloat16()
        y = x.clone()
        comm.all_reduce(y, two_shot=two)
        assert torch.equal(x, y)
    for T, H, S in ((20, 4096, 4), (3, 8192, 8), (64, 4096, 0)):
        res = torch.randn(T, H, device=dev).bfloat16()
        x = torch.randn(S, T, H, device=dev) if S else torch.randn(T, H, device=dev).bfloat16()
        r1, r2 = res.clone(), res.clone()
        ss1 = C.residual_reduce(x, r1, True)
        ss2 = comm.residual_allreduce(x, r2, True)
        assert torch.equal(r1, r2) and torch.allclose(ss1, ss2, rtol=1e-6)
    assert comm.errors() == 0 and comm.calls == 7
    comm.close()


def test_xgmi_collectives_two_rank_processes_share_one_gpu():
    """Two rank processes on this one GPU: buffers exported / mapped through HIP IPC, each
    rank's kernels spin on flags the other rank's kernels store — the protocol of the 8-GPU
    node minus the links.  Every result is checked against sums formed over gloo."""
    import json
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "xgmi_selftest.py"), "--world", "2", "--same-gpu", "--iters", "30"], capture_output=True, text=True,
                       timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["ok"] and all(g["timeouts"] == 0 for g in out["ranks"])


TP2_ONE_GPU_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["KV_ROOT"])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams
from kvmini.models.llama import get_config, random_weights
dist.init_process_group("gloo")  # host-side bootstrap: both ranks share GPU 0, which RCCL refuses
r, ws = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
MODEL = os.environ.get("KV_TP_MODEL", "small-hd128")
cfg = get_config(MODEL)
w = random_weights(cfg, "cuda:0", torch.bfloat16, seed=3, tp_rank=r, tp_size=ws)
eng = LLMEngine(EngineConfig(model=MODEL, device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, num_blocks=128, kernels="hip", tp_size=ws, tp_rank=r,
                             use_graphs=True), weights=w, tp_group=dist.group.WORLD)
eng.warmup()
assert eng.runner.xgmi is not None and eng.runner.graphs
prompts = [[1, 50, 60, 70, 80], list(range(1, 30)), [1, 9, 8, 7]]
if r == 0:
    outs = eng.generate(prompts, SamplingParams(max_tokens=10, ignore_eos=True))
    base = LLMEngine(EngineConfig(model=MODEL, device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, num_blocks=128, kernels="hip", use_graphs=False),
                     weights=random_weights(cfg, "cuda:0", torch.bfloat16, seed=3))
    want = base.generate(prompts, SamplingParams(max_tokens=10, ignore_eos=True))
    # the shards sum in a different order than one GEMM: equal lengths, equal first tokens
    ok = [len(o) for o in outs] == [len(o) for o in want] and all(a[:3] == b[:3] for a, b in zip(outs, want)) and eng.pipelined_steps > 0
    print("OUTS", outs, want, eng.pipelined_steps, flush=True)
    # sampled requests (seeded and not): the scheduling rank's picks reach the follower on the device,
    # so these steps are launched ahead as well; the seeded request repeats itself
    before, steps0 = eng.pipelined_steps, eng.stats.steps
    sp = SamplingParams(max_tokens=12, ignore_eos=True, temperature=0.8, top_p=0.9, seed=11)
    s1 = eng.generate([prompts[1], prompts[0]], sp)
    s2 = eng.generate([prompts[1]], sp)
    s3 = eng.generate([prompts[2]], SamplingParams(max_tokens=12, ignore_eos=True, temperature=0.8))
    ran, ahead = eng.stats.steps - steps0, eng.pipelined_steps - before
    ok = ok and s1[0] == s2[0] and len(s3[0]) == 12 and ahead >= ran - 6
    print("SAMPLED", s1, s2, s3, ran, ahead, flush=True)
    eng.release_followers()
else:
    ok = eng.follow() >= 10
ok = ok and eng.runner.xgmi.errors() == 0 and eng.runner.xgmi.calls % 2 == 0
print(f"RANK{r} OK={ok} calls={eng.runner.xgmi.calls}", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


@pytest.mark.parametrize("world,model", [(2, "small-hd128"), (4, "small-hd128-kv4")])
def test_tensor_parallel_two_rank_processes_share_one_gpu(tmp_path, world, model):
    """TP = 2 and TP = 4 end to end on one GPU: one process per rank, each with its shard of the
    weights on GPU 0; the all-reduce behind every row-parallel projection runs in the fused peer-memory
    kernel between the processes (HIP IPC), inside captured HIP graphs; the other ranks follow the
    step images from the host channel; steps are launched ahead."""
    import subprocess
    import sys

    script = tmp_path / "tp.py"
    script.write_text(TP2_ONE_GPU_SCRIPT)
    env = dict(os.environ, KV_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", KV_TP_MODEL=model)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(29639 + world), str(script)],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert all(f"RANK{k} OK=True" in r.stdout for k in range(world))


@pytest.mark.gpu
def test_seeded_requests_are_sampled_on_the_device_and_stay_launched_ahead():
    """A seeded request's variate for its k-th token is a function of (seed, k) that the host
    computes at launch time and the sampling kernel consumes: the same tokens with and without
    launch-ahead, alone or in company, and the steps are launched ahead.  A guided request honours
    its seed as well."""
    from kvmini.engine import guided as G
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.engine.tokenizer import get_tokenizer
    from kvmini.models.llama import get_config, random_weights

    w = lambda: random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=3)  # noqa: E731
    common = dict(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, num_blocks=128, kernels="hip")
    prompts = [[1, 50, 60, 70, 80], list(range(1, 30))]
    sp = SamplingParams(max_tokens=16, ignore_eos=True, temperature=0.8, top_p=0.9, top_k=40, seed=5)
    outs = {}
    for mode in (True, False):
        eng = LLMEngine(EngineConfig(**common, async_scheduling=mode), weights=w())
        eng.warmup()
        p0 = eng.pipelined_steps
        outs[mode] = eng.generate(prompts, sp)
        if mode:
            assert eng.pipelined_steps - p0 >= 14
            alone = eng.generate([prompts[1]], sp)[0]
            other = eng.generate([prompts[1]], SamplingParams(**{**vars(sp), "seed": 6}))[0]
            tok = get_tokenizer(eng.mcfg.vocab_size, 1, eng.mcfg.eos_token_id)
            eng.set_token_classes(G.token_classes(tok, eng.mcfg.vocab_size, eng.mcfg.eos_token_id))
            gp = SamplingParams(max_tokens=24, temperature=1.0, seed=9, guided="json", guided_start=(G.S_START, 0, 0))
            g1, g2 = eng.generate([prompts[0]], gp)[0], eng.generate([prompts[0]], gp)[0]
            assert g1 == g2 and isinstance(__import__("json").loads(tok.decode(g1)), (dict, list))
    assert outs[True] == outs[False] and alone == outs[True][1] and other != alone


@pytest.mark.gpu
def test_sampling_penalties_run_on_the_device_and_stay_launched_ahead():
    """Presence / frequency / repetition penalties are device ops in front of the pick: the same tokens with and
    without launch-ahead, steps launched ahead, a strong frequency penalty keeps a greedy reply from repeating
    itself, and a request without penalties next to a penalised one is untouched."""
    from collections import Counter

    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights

    w = lambda: random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=3)  # noqa: E731
    common = dict(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, num_blocks=128, kernels="hip")
    prompts = [[1, 50, 60, 70, 80], list(range(1, 30))]
    pen = SamplingParams(max_tokens=40, ignore_eos=True, frequency_penalty=2.0, presence_penalty=1.0, repetition_penalty=1.3)
    plain = SamplingParams(max_tokens=40, ignore_eos=True)
    outs = {}
    for mode in (True, False):
        eng = LLMEngine(EngineConfig(**common, async_scheduling=mode), weights=w())
        eng.warmup()
        p0 = eng.pipelined_steps
        outs[mode] = eng.generate(prompts + [prompts[0]], [pen, pen, plain])
        if mode:
            assert eng.pipelined_steps - p0 >= 35
            alone = eng.generate([prompts[0]], plain)[0]
    assert outs[True] == outs[False]
    assert outs[True][2] == alone and outs[True][0] != alone
    assert max(Counter(outs[True][0]).values()) <= 2 and max(Counter(alone).values()) >= 1


@pytest.mark.gpu
def test_logit_bias_and_min_tokens_on_the_graph_path():
    """logit_bias bans and pushes tokens, min_tokens holds the stop token back, in steps launched ahead of the
    read-back on captured graphs; launch-ahead and synchronous stepping give the same tokens."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights

    w = lambda: random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=3)  # noqa: E731
    common = dict(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=256, num_blocks=128, kernels="hip")
    prompt = [1, 50, 60, 70, 80]
    plain = SamplingParams(max_tokens=24, ignore_eos=True)
    outs = {}
    for mode in (True, False):
        eng = LLMEngine(EngineConfig(**common, async_scheduling=mode), weights=w())
        eng.warmup()
        alone = eng.generate([prompt], plain)[0]
        k = next(i for i in range(1, 12) if alone[i] not in alone[:i])
        ban = SamplingParams(max_tokens=24, ignore_eos=True, logit_bias={alone[0]: -100.0, alone[1]: -100.0})
        push = SamplingParams(max_tokens=24, ignore_eos=True, logit_bias={777: 100.0})
        stop = SamplingParams(max_tokens=24, ignore_eos=True, stop_token_ids=[alone[k]])
        held = SamplingParams(max_tokens=24, ignore_eos=True, stop_token_ids=[alone[k]], min_tokens=16)
        outs[mode] = eng.generate([prompt] * 5, [ban, push, stop, held, plain])
        o = outs[mode]
        assert alone[0] not in o[0] and alone[1] not in o[0]
        assert o[1] == [777] * 24
        assert o[2] == alone[: k + 1]
        assert len(o[3]) >= 16 and alone[k] not in o[3][:16] and o[3][:k] == alone[:k]
        assert o[4] == alone
    assert outs[True] == outs[False]


# ---------------------------------------------------------------- FP8 (e4m3) KV cache
def _fp8_cache(nb, nkv, bs, hd, seed):
    g = torch.Generator(device=_dev()); g.manual_seed(seed)
    return (torch.randn(nb, nkv, bs, hd, device=_dev(), generator=g) * 1.5).to(torch.float8_e4m3fn)


@pytest.mark.gpu
def test_fp8_kv_write_matches_torch_quantisation():
    """rope_kv_write into an e4m3 cache: the bytes are those of torch's float8_e4m3fn conversion of
    x / scale (per kv head), clamped to +-448; values far outside the range saturate."""
    from kvmini.models.llama import get_config, rope_tables
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(3)
    nh, nkv, hd, bs, nb, T = 8, 2, 128, 16, 32, 37
    cos, sin = rope_tables(get_config("llama-3-8b"), 512, _dev())
    qkv = torch.randn(T, (nh + 2 * nkv) * hd, device=_dev()).bfloat16() * 3
    qkv[3, -hd:] = 2000.0  # saturates
    qkv[4, -hd:] = -1e-4   # below the smallest subnormal: rounds to (minus) zero
    pos = torch.randint(0, 512, (T,), device=_dev(), dtype=torch.int32)
    slots = torch.randperm(nb * bs, device=_dev())[:T].int()
    slots[5] = -1
    sc = torch.tensor([[0.5, 2.0], [1.0, 0.25]], device=_dev())
    kc, vc = _fp8_cache(nb, nkv, bs, hd, 1), _fp8_cache(nb, nkv, bs, hd, 2)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    q_ref = torch_ops.rope_kv_write(qkv.clone(), pos.long(), cos, sin, kc_ref, vc_ref, slots.long(), nh, nkv, hd, kv_scale=sc)
    q = hip_ops.rope_kv_write(qkv.clone(), pos, cos, sin, kc, vc, slots, nh, nkv, hd, kv_scale=sc)
    assert (q.float() - q_ref.float()).abs().max().item() <= 3e-2
    assert torch.equal(vc.view(torch.uint8), vc_ref.view(torch.uint8))  # V is not rotated: byte for byte
    # K goes through RoPE in fp32 on both sides but is rounded to bf16 in between by the kernel: allow one e4m3 step
    dk = (kc.float() - kc_ref.float()).abs()
    assert (dk <= 0.13 * kc_ref.float().abs() + 2e-3).all()
    assert (kc.view(torch.uint8) == kc_ref.view(torch.uint8)).float().mean().item() > 0.97
    assert vc.float().abs().max().item() == 448.0


@pytest.mark.gpu
@pytest.mark.parametrize("rep", [1, 4, 8])
@pytest.mark.parametrize("hd", [128, 64])
def test_fp8_kv_paged_attention_matches_fp32_reference(rep, hd):
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(4)
    nkv, bs, nb = 2, 16, 64
    nh = nkv * rep
    n_seqs, max_blocks, T = 4, 12, 9
    kc, vc = _fp8_cache(nb, nkv, bs, hd, 1), _fp8_cache(nb, nkv, bs, hd, 2)
    sc = torch.tensor([[0.7, 1.3], [2.0, 0.5]], device=_dev())
    bt = torch.randperm(nb, device=_dev())[: n_seqs * max_blocks].view(n_seqs, max_blocks).int()
    tok_seq = torch.randint(0, n_seqs, (T,), device=_dev(), dtype=torch.int32)
    ctx = torch.tensor([1, 5, 16, 17, 33, 100, 190, 0, 64], device=_dev(), dtype=torch.int32)
    q = torch.randn(T, nh, hd, device=_dev()).bfloat16()
    scale = 1 / math.sqrt(hd)
    ref = torch_ops.paged_attention(q, kc, vc, bt, tok_seq, ctx, scale, kv_scale=sc)
    out = hip_ops.paged_attention(q, kc, vc, bt, tok_seq, ctx, scale, kv_scale=sc)
    assert (out.float() - ref.float()).abs().max().item() <= 3e-2
    assert out[7].abs().max().item() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kv8", [False, True])
@pytest.mark.parametrize("rep, hd", [(4, 128), (8, 128), (1, 128), (4, 64)])
def test_paged_attention_long_walks_every_remainder(kv8, rep, hd):
    """The decode kernel walks a sequence's blocks in rounds of three register stages per wave: contexts whose
    blocks per wave are 3 n, 3 n + 1 and 3 n + 2 (several rounds, so the loop's back edge runs), odd block counts
    over the waves, and short ones next to them in one launch."""
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(9)
    nkv, bs = 2, 16
    nh = nkv * rep
    ctx_list = [193, 208, 209, 250, 385, 400, 577, 640, 777, 1000, 1023, 1024, 3, 0, 49]
    T = len(ctx_list)
    max_blocks = 64
    nb = T * max_blocks
    if kv8:
        kc, vc = _fp8_cache(nb, nkv, bs, hd, 1), _fp8_cache(nb, nkv, bs, hd, 2)
        sc = torch.tensor([[0.7, 1.3], [2.0, 0.5]], device=_dev())
    else:
        kc, vc = _rand_cache(nb, nkv, bs, hd), _rand_cache(nb, nkv, bs, hd)
        sc = None
    bt = torch.randperm(nb, device=_dev()).view(T, max_blocks).int()
    tok_seq = torch.arange(T, device=_dev(), dtype=torch.int32)
    ctx = torch.tensor(ctx_list, device=_dev(), dtype=torch.int32)
    q = torch.randn(T, nh, hd, device=_dev()).bfloat16()
    scale = 1 / math.sqrt(hd)
    kw = dict(kv_scale=sc) if kv8 else {}
    ref = torch_ops.paged_attention(q, kc, vc, bt, tok_seq, ctx, scale, **kw)
    out = hip_ops.paged_attention(q, kc, vc, bt, tok_seq, ctx, scale, **kw)
    err = (out.float() - ref.float()).abs().amax(dim=(1, 2)) if out.dim() == 3 else (out.float() - ref.float()).abs().view(T, -1).amax(dim=1)
    assert err.max().item() <= 2e-2, err.tolist()
    assert out[13].abs().max().item() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("rep,nkv", [(1, 2), (4, 2), (8, 2), (4, 8)])
def test_fp8_kv_prefill_attention_matches_fp32_reference(rep, nkv):
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(11)
    hd, bs, nb = 128, 16, 200
    nh = nkv * rep
    n_seqs, max_blocks = 4, 40
    chunks = [(0, 0, 600), (1, 37, 250), (2, 0, 5), (3, 300, 200)] if nkv == 8 else [(0, 0, 40), (1, 37, 23), (2, 0, 5), (3, 300, 33)]
    kc, vc = _fp8_cache(nb, nkv, bs, hd, 1), _fp8_cache(nb, nkv, bs, hd, 2)
    sc = (torch.rand(2, nkv, device=_dev()) + 0.5).contiguous()
    bt = torch.randperm(nb, device=_dev())[: n_seqs * max_blocks].view(n_seqs, max_blocks).int()
    rows, tok_seq, ctx, tiles = 0, [], [], []
    for seq, p0, n in chunks:
        for off in range(0, n, 16):
            tiles.append((rows + off, min(16, n - off), p0 + off, seq))
        tok_seq += [seq] * n
        ctx += [p0 + j + 1 for j in range(n)]
        rows += n
    q = torch.randn(rows, nh, hd, device=_dev()).bfloat16()
    tok_seq_t = torch.tensor(tok_seq, device=_dev(), dtype=torch.int32)
    ctx_t = torch.tensor(ctx, device=_dev(), dtype=torch.int32)
    scale = 1 / math.sqrt(hd)
    ref = torch_ops.paged_attention(q, kc, vc, bt, tok_seq_t, ctx_t, scale, kv_scale=sc)
    out = torch.zeros((rows, nh * hd), device=_dev(), dtype=torch.bfloat16)
    hip_ops.prefill_attention(q, kc, vc, bt, torch.tensor(tiles, device=_dev(), dtype=torch.int32).T.contiguous(), out, scale, kv_scale=sc)
    assert (out.float() - ref.float()).abs().max().item() <= 4e-2


@pytest.mark.gpu
@pytest.mark.parametrize("rep", [1, 4, 8])
def test_fp8_kv_fused_decode_attention_matches_two_kernel_path(rep):
    """One launch (RoPE + e4m3 KV write + attention, the token's own key / value taken as the cache
    will hold them) == rope_kv_write followed by paged_attention on the FP8 cache."""
    from kvmini.models.llama import LlamaConfig, rope_tables
    from kvmini.ops import hip_ops

    torch.manual_seed(7)
    nkv, hd, bs, nb = 2, 128, 16, 96
    nh = nkv * rep
    n_seqs, max_blocks, T, S, h_in = 16, 6, 16, 5, 512
    width = (nh + 2 * nkv) * hd
    cos, sin = rope_tables(LlamaConfig(head_dim=hd), 128, _dev())
    ctx = torch.tensor([1, 2, 16, 17, 33, 64, 80, 0, 5, 48, 96, 31, 32, 0, 9, 65], device=_dev(), dtype=torch.int32)
    positions = (ctx - 1).clamp(min=0)
    tok_seq = torch.arange(T, device=_dev(), dtype=torch.int32)
    bt = torch.randperm(nb, device=_dev())[: n_seqs * max_blocks].view(n_seqs, max_blocks).int()
    slots = torch.where(ctx > 0, bt[torch.arange(T, device=_dev()), (positions // bs).long()] * bs + positions % bs, torch.full_like(ctx, -1)).int()
    slabs = torch.randn(S, T, width, device=_dev()) * 0.4
    ss = torch.rand(T, 1, device=_dev()) * h_in + 1.0
    sc = torch.tensor([[0.5, 1.5], [1.25, 0.75]], device=_dev())
    part = hip_ops.Partial(slabs, ss, 1e-5, h_in)
    kc1, vc1 = _fp8_cache(nb, nkv, bs, hd, 1), _fp8_cache(nb, nkv, bs, hd, 2)
    q = hip_ops.rope_kv_write(part, positions, cos, sin, kc1, vc1, slots, nh, nkv, hd, kv_scale=sc)
    ref = hip_ops.paged_attention(q, kc1, vc1, bt, tok_seq, ctx, 1 / math.sqrt(hd), kv_scale=sc)
    kc2, vc2 = _fp8_cache(nb, nkv, bs, hd, 1), _fp8_cache(nb, nkv, bs, hd, 2)
    out = hip_ops.rope_attention_fused(part, positions, cos, sin, kc2, vc2, slots, bt, tok_seq, ctx, nh, 1 / math.sqrt(hd), kv_scale=sc)
    assert torch.equal(kc1.view(torch.uint8), kc2.view(torch.uint8)) and torch.equal(vc1.view(torch.uint8), vc2.view(torch.uint8))
    assert (out.float() - ref.float()).abs().max().item() <= 2e-2
    assert out[7].abs().max().item() == 0 and out[13].abs().max().item() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("graphs", [False, True])
def test_engine_fp8_kv_cache_tracks_the_bf16_engine(graphs):
    """kv_cache_dtype=fp8 end to end (prompt chunks on the tiled kernel, decode on the fused one,
    prefix-cache reuse): the logits stay close to the bf16-cache engine's."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights

    common = dict(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=64, max_model_len=512, kernels="hip", use_graphs=graphs, kv_memory_fraction=0.01)
    w = lambda: random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=3)  # noqa: E731
    base = LLMEngine(EngineConfig(**common), weights=w())
    eng = LLMEngine(EngineConfig(**common, kv_cache_dtype="fp8"), weights=w())
    assert eng.runner.kv_cache.dtype == torch.float8_e4m3fn and eng.runner.kv_cache.element_size() == 1
    if graphs:
        base.warmup(); eng.warmup()
    prompts = [list(range(1, 100)), [1, 50, 60, 70, 80], list(range(5, 45))]
    params = SamplingParams(max_tokens=12, ignore_eos=True)
    want, got = base.generate(prompts, params), eng.generate(prompts, params)
    agree = sum(a == b for x, y in zip(want, got) for a, b in zip(x, y)) / sum(len(x) for x in want)
    assert agree >= 0.6, (agree, want, got)  # random weights: near-ties flip under e4m3 rounding, most tokens must survive
    assert got[0][0] == want[0][0]
    # logits of one prefill step, same inputs
    for e in (base, eng):
        e.scheduler.add(e.make_sequence(list(range(1, 60)), SamplingParams(max_tokens=1, logprobs=1)))
    lb = base.runner.execute(base.scheduler.schedule())[1].float()
    le = eng.runner.execute(eng.scheduler.schedule())[1].float()
    rel = (lb - le).norm() / lb.norm()
    assert rel.item() < 0.08, rel.item()


@pytest.mark.gpu
@pytest.mark.parametrize("rep,kv8", [(4, False), (8, False), (1, False), (4, True), (8, True)])
def test_split_kv_decode_attention_matches_the_unsplit_kernel(rep, kv8):
    """Several workgroups per (token, kv head), merged by the last to arrive, against one workgroup
    per pair: contexts from one token (one split does everything) to 3000 tokens (12 splits of 16
    blocks), a padding row, repeated launches on the same workspace (counters reset themselves)."""
    from kvmini.models.llama import LlamaConfig, rope_tables
    from kvmini.ops import hip_ops

    torch.manual_seed(9)
    nkv, hd, bs = 2, 128, 16
    nh = nkv * rep
    ctx = torch.tensor([1, 2, 255, 257, 600, 0, 3000, 1024], device=_dev(), dtype=torch.int32)
    T = ctx.numel()
    max_blocks = 3008 // bs
    nb = T * max_blocks + 8
    width = (nh + 2 * nkv) * hd
    cos, sin = rope_tables(LlamaConfig(head_dim=hd), 3072, _dev())
    positions = (ctx - 1).clamp(min=0)
    tok_seq = torch.arange(T, device=_dev(), dtype=torch.int32)
    bt = torch.randperm(nb, device=_dev())[: T * max_blocks].view(T, max_blocks).int()
    slots = torch.where(ctx > 0, bt[torch.arange(T, device=_dev()), (positions // bs).long()] * bs + positions % bs, torch.full_like(ctx, -1)).int()
    slabs = torch.randn(3, T, width, device=_dev()) * 0.4
    part = hip_ops.Partial(slabs, torch.rand(T, 1, device=_dev()) * 512 + 1.0, 1e-5, 512)
    sc = torch.tensor([[0.5, 1.5], [1.25, 0.75]], device=_dev()) if kv8 else None

    def caches():
        if kv8:
            return _fp8_cache(nb, nkv, bs, hd, 1), _fp8_cache(nb, nkv, bs, hd, 2)
        g = torch.Generator(device=_dev()); g.manual_seed(1)
        return (torch.randn(nb, nkv, bs, hd, device=_dev(), generator=g).bfloat16(), torch.randn(nb, nkv, bs, hd, device=_dev(), generator=g).bfloat16())

    kc1, vc1 = caches()
    ref = hip_ops.rope_attention_fused(part, positions, cos, sin, kc1, vc1, slots, bt, tok_seq, ctx, nh, 1 / math.sqrt(hd), kv_scale=sc, n_split=1)
    for n_split in (2, 12, 32):
        for _ in range(3):
            kc2, vc2 = caches()
            out = hip_ops.rope_attention_fused(part, positions, cos, sin, kc2, vc2, slots, bt, tok_seq, ctx, nh, 1 / math.sqrt(hd), kv_scale=sc, n_split=n_split)
            assert torch.equal(kc1.view(torch.uint8), kc2.view(torch.uint8)) and torch.equal(vc1.view(torch.uint8), vc2.view(torch.uint8))
            assert (out.float() - ref.float()).abs().max().item() <= 8e-3, n_split
            assert out[5].abs().max().item() == 0
    torch.cuda.synchronize()
    assert int(hip_ops._split_ws[_dev()][:4096].view(torch.int32).abs().sum()) == 0  # every arrival counter is back at zero
    assert hip_ops.decode_attention_splits(1, 8) == 32 and hip_ops.decode_attention_splits(16, 8) == 32 and hip_ops.decode_attention_splits(32, 8) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_engine_long_context_decode_takes_the_split_kv_graph_and_matches_eager(kv):
    """A 3000-token sequence decoded alone: the engine replays the split-KV variant of the
    decode-only graph (context >= 2048 tokens on a 16-row step) and produces the tokens of the
    eager, unsplit path; a short request next to it still runs on the plain variant."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights

    common = dict(model="small-hd128", device="cuda:0", max_num_seqs=4, max_batched_tokens=512, max_model_len=4096, kernels="hip", kv_cache_dtype=kv, kv_memory_fraction=0.02)
    w = lambda: random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=3)  # noqa: E731
    prompts = [[1] + [(7 * i) % 900 + 10 for i in range(2999)]]
    params = SamplingParams(max_tokens=24, ignore_eos=True)
    os.environ["KVMINI_DECODE_SPLITS"] = "1"
    try:
        want = LLMEngine(EngineConfig(**common, use_graphs=False), weights=w()).generate(prompts, params)
    finally:
        del os.environ["KVMINI_DECODE_SPLITS"]
    eng = LLMEngine(EngineConfig(**common), weights=w())
    eng.warmup()
    assert (16, 2) in eng.runner.graphs and (16, 1) in eng.runner.graphs
    assert eng.runner.step_kind(16, True, 3000) == 2 and eng.runner.step_kind(16, True, 500) == 1 and eng.runner.step_kind(16, False, 3000) == 0
    got = eng.generate(prompts, params)
    agree = sum(a == b for a, b in zip(want[0], got[0])) / len(want[0])
    assert agree >= 0.9, (agree, want, got)  # fp32 partial sums are merged in another order: a near-tie may flip late in the sequence
    assert got[0][:8] == want[0][:8]
    short = eng.generate([[1, 5, 9, 11]], SamplingParams(max_tokens=8, ignore_eos=True))
    assert len(short[0]) == 8


@pytest.mark.gpu
@pytest.mark.parametrize("rep,kv8", [(4, False), (8, False), (2, False), (4, True)])
def test_split_kv_attention_over_listed_rows_matches_the_fp32_reference(rep, kv8):
    """Decode tokens of a step that also carries prompt chunks: listed rows (any rows of a wide
    step, -1 entries unused) against fp32 paged attention over the same cache; contexts from one
    token to 5000; rows outside the list keep their content; repeated launches on one workspace."""
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(13)
    nkv, hd, bs = 2, 128, 16
    nh = nkv * rep
    T = 300
    live = [0, 7, 64, 129, 255, 299, 33]
    ctx_of = [1, 17, 5000, 256, 2049, 700, 4097]
    ctx = torch.full((T,), -1, device=_dev(), dtype=torch.int32)  # prompt rows: not this kernel's
    ctx[torch.tensor(live, device=_dev())] = torch.tensor(ctx_of, device=_dev(), dtype=torch.int32)
    n_seq = len(live)
    max_blocks = 5008 // bs
    nb = n_seq * max_blocks + 8
    tok_seq = torch.zeros(T, device=_dev(), dtype=torch.int32)
    tok_seq[torch.tensor(live, device=_dev())] = torch.arange(n_seq, device=_dev(), dtype=torch.int32)
    bt = torch.randperm(nb, device=_dev())[: n_seq * max_blocks].view(n_seq, max_blocks).int()
    sc = torch.tensor([[0.5, 1.5], [1.25, 0.75]], device=_dev()) if kv8 else None
    if kv8:
        kc, vc = _fp8_cache(nb, nkv, bs, hd, 1), _fp8_cache(nb, nkv, bs, hd, 2)
    else:
        kc, vc = torch.randn(nb, nkv, bs, hd, device=_dev()).bfloat16(), torch.randn(nb, nkv, bs, hd, device=_dev()).bfloat16()
    qkv = torch.randn(T, (nh + 2 * nkv) * hd, device=_dev()).bfloat16()
    q = qkv[:, : nh * hd].view(T, nh, hd)  # strided rows, as rope_kv_write returns them
    scale = 1 / math.sqrt(hd)
    ref = torch_ops.paged_attention(q.float(), kc, vc, bt, tok_seq, ctx.clamp(min=0), scale, max_blocks, sc)
    rows = torch.full((16,), -1, device=_dev(), dtype=torch.int32)
    order = [4, 0, 2, 6, 1, 5, 3]
    rows[torch.tensor([0, 1, 2, 5, 6, 9, 15], device=_dev())] = torch.tensor([live[i] for i in order], device=_dev(), dtype=torch.int32)
    for n_split in (None, 1, 5, 32):
        for _ in range(2):
            out = torch.full((T, nh * hd), 3.0, device=_dev(), dtype=torch.bfloat16)
            hip_ops.split_decode_attention(q, kc, vc, bt, tok_seq, ctx, rows, out, scale, kv_scale=sc, n_split=n_split)
            lv = torch.tensor(live, device=_dev())
            err = (out[lv].float() - ref[lv].float()).abs().max().item()
            assert err <= (3e-2 if kv8 else 1.2e-2), (n_split, err)
            rest = torch.ones(T, dtype=torch.bool, device=_dev()); rest[lv] = False
            assert bool((out[rest] == 3.0).all())
    torch.cuda.synchronize()
    assert int(hip_ops._split_ws[_dev()][:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_engine_decodes_ride_along_with_a_prompt_on_the_split_kv_kernel(kv, monkeypatch):
    """A 3000-token sequence keeps decoding while another long prompt is fed in chunks: those
    steps list the decode row for the split-KV kernel (threshold lowered to 2048 tokens here) and
    the tokens equal those of the per-token kernel (threshold out of reach)."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights
    from kvmini.ops import hip_ops

    common = dict(model="small-hd128", device="cuda:0", max_num_seqs=4, max_batched_tokens=256, max_model_len=4096, kernels="hip", kv_cache_dtype=kv, kv_memory_fraction=0.02)
    w = lambda: random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=3)  # noqa: E731
    first = [1] + [(7 * i) % 900 + 10 for i in range(2999)]
    second = [1] + [(11 * i) % 900 + 10 for i in range(1499)]

    def run(min_ctx):
        eng = LLMEngine(EngineConfig(**common), weights=w())
        eng.runner.DEC_SPLIT_MIN_CTX = min_ctx
        eng.warmup()
        calls = []
        real = hip_ops.split_decode_attention
        monkeypatch.setattr(hip_ops, "split_decode_attention", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        s1 = eng.make_sequence(first, SamplingParams(max_tokens=40, ignore_eos=True))
        eng.scheduler.add(s1)
        while len(s1.output_ids) < 4:
            eng.step()
        s2 = eng.make_sequence(second, SamplingParams(max_tokens=8, ignore_eos=True))
        eng.scheduler.add(s2)  # 1500 tokens at 256 per step: six steps that also carry s1's decode token
        while eng.scheduler.has_work():
            eng.step()
        monkeypatch.setattr(hip_ops, "split_decode_attention", real)
        return list(s1.output_ids), list(s2.output_ids), len(calls), eng.runner.variant_graph_steps

    a1, a2, n_a, replays = run(2048)
    b1, b2, n_b, _ = run(1 << 30)
    layers = get_config("small-hd128").num_layers
    # the first step of the shape runs eagerly, the second is captured (both call the op once per layer), later ones replay the graph
    assert n_a >= 2 * layers and n_a + layers * replays >= 5 * layers and replays >= 3 and n_b == 0, (n_a, replays, n_b)
    agree = sum(x == y for x, y in zip(a1, b1)) / len(b1)
    assert agree >= 0.9 and a1[:8] == b1[:8], (agree, a1, b1)
    assert a2 == b2 or sum(x == y for x, y in zip(a2, b2)) >= 7


@pytest.mark.gpu
def test_short_chunks_deep_in_a_context_are_served_as_listed_rows(monkeypatch):
    """Speculative verification (three positions per step) 3000 tokens into a sequence: the chunk is
    listed row by row for the split-KV kernel (each row with its own causal length) instead of one
    tile whose workgroup walks the context alone; tokens equal those of the tiled path."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights
    from kvmini.ops import hip_ops

    common = dict(model="small-hd128", device="cuda:0", max_num_seqs=4, max_batched_tokens=512, max_model_len=4096, kernels="hip", kv_memory_fraction=0.02,
                  draft_model="small-hd128-draft", num_speculative_tokens=2)
    w = lambda: random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=3)  # noqa: E731
    prompts = [[1] + [(7 * i) % 900 + 10 for i in range(2999)]]
    params = SamplingParams(max_tokens=24, ignore_eos=True)

    def run(min_ctx):
        eng = LLMEngine(EngineConfig(**common), weights=w())
        eng.runner.DEC_SPLIT_MIN_CTX = min_ctx
        eng.warmup()
        calls = []
        real = hip_ops.split_decode_attention
        monkeypatch.setattr(hip_ops, "split_decode_attention", lambda *a, **k: (calls.append(int(a[6].numel())), real(*a, **k))[1])
        out = eng.generate(prompts, params)
        monkeypatch.setattr(hip_ops, "split_decode_attention", real)
        return out[0], calls, eng.stats.spec_proposed

    a, calls_a, prop = run(2048)
    b, calls_b, _ = run(1 << 30)
    assert prop > 0 and calls_a and not calls_b
    assert set(calls_a) <= {1, 2, 4}, set(calls_a)  # three rows per verification, rounded up to four
    agree = sum(x == y for x, y in zip(a, b)) / len(b)
    assert len(a) == len(b) and agree >= 0.9 and a[:8] == b[:8], (agree, a, b)


@pytest.mark.gpu
def test_decode_only_graphs_of_wide_buckets_match_eager():
    """100 sequences decoding together (the 128-token bucket: 96-row and 128-row forms of the streaming GEMM, the
    decode-only graph without fused attention) produce the tokens of the eager path."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights

    common = dict(model="small-hd128", device="cuda:0", max_num_seqs=128, max_batched_tokens=512, max_model_len=256, kernels="hip", kv_memory_fraction=0.02)
    w = lambda: random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=5)  # noqa: E731
    prompts = [[1] + [(13 * i + 7 * j) % 900 + 10 for j in range(3 + i % 9)] for i in range(100)]
    params = SamplingParams(max_tokens=6, ignore_eos=True)
    want = LLMEngine(EngineConfig(**common, use_graphs=False), weights=w()).generate(prompts, params)
    eng = LLMEngine(EngineConfig(**common), weights=w())
    eng.warmup()
    assert (128, 1) in eng.runner.graphs and (96, 1) in eng.runner.graphs and (64, 1) in eng.runner.graphs
    got = eng.generate(prompts, params)
    same = sum(a == b for a, b in zip(want, got))
    assert same >= 97, (same, [(a, b) for a, b in zip(want, got) if a != b][:3])  # (other GEMM forms per bucket: a near-tie may flip)
    assert all(a[:2] == b[:2] for a, b in zip(want, got))


@pytest.mark.gpu
@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_hundred_sequences_with_longer_contexts_agree_with_small_batches(kv):
    """100 sequences of 150-230 prompt tokens decode together (rows beyond the fused kernel's 64: RoPE / cache write and
    the batch decode-attention kernel — MFMA scores, three rotating K/V stages, several blocks per wave) and the same
    prompts four at a time (the fused decode kernel): the first tokens agree everywhere, whole replies nearly
    everywhere (different kernels round differently; a near-tie may flip)."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights

    w = lambda: random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=5)  # noqa: E731
    prompts = [[1] + [(13 * i + 7 * j) % 900 + 10 for j in range(150 + (i * 37) % 80)] for i in range(100)]
    params = SamplingParams(max_tokens=8, ignore_eos=True)
    common = dict(model="small-hd128", device="cuda:0", max_batched_tokens=2048, max_model_len=512, kernels="hip", kv_cache_dtype=kv, kv_memory_fraction=0.05)
    small = LLMEngine(EngineConfig(**common, max_num_seqs=4), weights=w())
    small.warmup()
    want = small.generate(prompts, params)
    del small
    big = LLMEngine(EngineConfig(**common, max_num_seqs=128), weights=w())
    big.warmup()
    got = big.generate(prompts, params)
    assert all(len(g) == 8 for g in got)
    assert all(a[0] == b[0] for a, b in zip(want, got))
    same = sum(a == b for a, b in zip(want, got))
    assert same >= 90, (same, [(a, b) for a, b in zip(want, got) if a != b][:3])


@pytest.mark.gpu
@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_new_turns_behind_a_cached_prefix_take_the_split_tile_kernel_inside_a_graph(monkeypatch, kv):
    """Chat continuation: a 3000-token conversation is in the prefix cache and every request adds a new turn of 40
    tokens.  The turn's tiles go to the split form of the tile kernel (key ranges shared out over the chip), from the
    second step of that shape on inside a captured graph; tokens equal those of the unsplit, eager path."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights
    from kvmini.ops import hip_ops

    common = dict(model="small-hd128", device="cuda:0", max_num_seqs=4, max_batched_tokens=512, max_model_len=4096, kernels="hip", kv_cache_dtype=kv, kv_memory_fraction=0.02)
    w = lambda: random_weights(get_config("small-hd128"), "cuda:0", torch.bfloat16, seed=3)  # noqa: E731
    history = [1] + [(7 * i) % 900 + 10 for i in range(2999)]
    turns = [history + [(11 * i + 37 * t) % 900 + 10 for i in range(40)] for t in range(4)]
    params = SamplingParams(max_tokens=6, ignore_eos=True)

    def run(split):
        eng = LLMEngine(EngineConfig(**common), weights=w())
        eng.runner.PF_SPLIT_MIN_CTX = 2048 if split else 1 << 30
        eng.warmup()
        eng.generate([history], SamplingParams(max_tokens=1, ignore_eos=True))  # fills the prefix cache
        calls = []
        real = hip_ops.prefill_attention_split
        monkeypatch.setattr(hip_ops, "prefill_attention_split", lambda *a, **k: (calls.append((int(a[5]), int(a[8]))), real(*a, **k))[1])
        outs = [eng.generate([t], params)[0] for t in turns]
        monkeypatch.setattr(hip_ops, "prefill_attention_split", real)
        return outs, calls, eng.runner.variant_graph_steps, eng.bm.prefix_hits

    a, calls_a, graph_steps, hits = run(True)
    b, calls_b, _, _ = run(False)
    assert hits > 0 and calls_a and not calls_b
    assert all(tiles == 4 and shares >= 2 for tiles, shares in calls_a), calls_a  # 40 + a few re-computed tokens: 3-4 tiles, rounded up to 4
    assert graph_steps >= 2, graph_steps  # first turn eager, second captures, the rest replay
    for x, y in zip(a, b):
        assert len(x) == len(y) and x[:3] == y[:3] and sum(p == q for p, q in zip(x, y)) >= len(y) - 1, (a, b)


# ---------------------------------------------------------------- MXFP4 (4-bit) weights
def _fp8_rows(x):
    """x as wide steps of quantised models see it: per-token dynamic e4m3 (scale = absmax / 448), dequantised."""
    from kvmini.ops import hip_ops

    _x8, sc = hip_ops.native().quantize_rows_fp8(x, False)
    return (x.float() / sc[:, None]).to(torch.float8_e4m3fn).float() * sc[:, None]


@pytest.mark.gpu
@pytest.mark.parametrize("M", [16, 32, 64, 20, 100, 128, 300])
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 4096), (4096, 14336), (256, 256), (1000, 512)])
def test_stream_gemm_mxfp4_weights(M, N, K):
    """y = x @ dequant(W_mxfp4)^T against an fp32 matmul with the SAME quantised weights (the
    tolerance covers accumulation order only): nibble order, E8M0 scales per (row, 32 columns),
    every row-count path (16..64 rows, padded rows, the 128-row pass, several passes), slabs."""
    from kvmini.models.llama import dequantize_mxfp4, quantize_mxfp4
    from kvmini.ops import hip_ops

    torch.manual_seed(5)
    W = (torch.randn(N, K, device=_dev()) * 0.05).bfloat16()
    W[3] *= 37.0           # blocks with very different scales
    W[:, 64:96] *= 1e-3
    W[5, 32:64] = 0.0      # an all-zero block
    x = torch.randn(M, K, device=_dev()).bfloat16()
    codes, scales = quantize_mxfp4(W)
    Wdq = dequantize_mxfp4(codes, scales, torch.float32)
    # (more than 64 rows with a width the tiled kernel takes: MXFP4 x FP8 on the scaled MFMA — the activations are e4m3 there)
    wide = M > 64 and K % 128 == 0 and ((N + 63) // 64 * 64) % 128 == 0
    ref = (_fp8_rows(x) if wide else x.float()) @ Wdq.T
    tol = 1e-2 * ref.abs().max().item() + 1e-2
    pw = hip_ops.PackedLinear(W, mxfp4=True)
    assert pw.bits == 4 and pw.wp.dtype == torch.uint8 and pw.wp.numel() * 2 == pw.n_pad * K and pw.wscale.numel() * 128 == pw.n_pad * K
    y = hip_ops.linear(x, pw)
    assert y.shape == (M, N)
    assert (y.float() - ref).abs().max().item() <= tol
    s = hip_ops.linear(x, pw, allow_slabs=True)
    got = s.slabs.sum(0) if isinstance(s, hip_ops.Partial) else s.float()
    assert (got - ref).abs().max().item() <= tol
    # the quantisation error itself: e2m1 has one mantissa bit (relative step up to 1/3 inside a block)
    full = x.float() @ W.float().T
    assert ((ref - full).norm() / full.norm()).item() < 0.2
    if wide:  # KVMINI_FP8_ACT=0: bf16 activations on the exact bf16 expansion of the matrix
        prev, hip_ops.FP8_ACTIVATIONS = hip_ops.FP8_ACTIVATIONS, False
        try:
            y = hip_ops.linear(x, pw)
        finally:
            hip_ops.FP8_ACTIVATIONS = prev
        ref16 = x.float() @ Wdq.T
        assert (y.float() - ref16).abs().max().item() <= 1e-2 * ref16.abs().max().item() + 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("M,bits,nw,nb", [(128, 16, 4, 2), (128, 16, 8, 2), (256, 16, 4, 2), (128, 8, 4, 2), (128, 8, 8, 2), (256, 8, 4, 2), (128, 4, 4, 2), (128, 4, 8, 2),
                                          (16, 4, 4, 8), (32, 4, 8, 4), (48, 4, 4, 2), (64, 4, 8, 4)])
def test_stream_gemm3_two_feature_tiles_per_wave(C, M, bits, nw, nb):
    """ft = 2 (a wave owns two consecutive 16-feature tiles that share every x fragment read):
    bf16 output, split-K slabs and the SwiGLU epilogue against an fp32 matmul on the same
    (quantised) weights, for every weight format and both workgroup sizes."""
    from kvmini.models.llama import dequantize_mxfp4, quantize_fp8_rows, quantize_mxfp4
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(M + bits + nw)
    N, K = 16 * nw * 2 * 3, 2048
    W = (torch.randn(N, K, device=_dev()) * 0.05).bfloat16()
    x = torch.randn(M, K, device=_dev()).bfloat16()
    if bits == 4:
        Wd = dequantize_mxfp4(*quantize_mxfp4(W), torch.float32)
    elif bits == 8:
        q, sc = quantize_fp8_rows(W)
        Wd = q.float() * sc[:, None]
    else:
        Wd = W.float()
    pw = hip_ops.PackedLinear(W, fp8=bits == 8, mxfp4=bits == 4)
    wb = 4 if bits == 4 else 0
    ref = x.float() @ Wd.T
    tol = 1e-2 * ref.abs().max().item() + 1e-2
    y = torch.full((M, N), float("nan"), device=_dev(), dtype=torch.bfloat16)
    C.stream_gemm3(pw.wp, N, K, x, y, 0, nb, 1, K // (128 * nb), None, 0.0, pw.wscale, nw, wb, ft=2)
    assert (y.float() - ref).abs().max().item() <= tol
    ngroups = K // (128 * nb)
    for splits in (2, 3):  # 3: the last split is shorter
        gps = -(-ngroups // splits)
        n_s = -(-ngroups // gps)
        slabs = torch.full((n_s, M, N), float("nan"), device=_dev(), dtype=torch.float32)
        C.stream_gemm3(pw.wp, N, K, x, slabs, 1, nb, n_s, gps, None, 0.0, pw.wscale, nw, wb, ft=2)
        err = (slabs.sum(0) - ref).abs().max().item()
        assert err <= (2e-3 if bits != 8 else 1e-2) * ref.abs().max().item() + 1e-3, (splits, err)  # fp32 slabs (FP8: the row scale is applied per slab)
    ss = (x.float() ** 2).view(M, -1, 512).sum(-1).contiguous()
    act = torch.full((M, N // 2), float("nan"), device=_dev(), dtype=torch.bfloat16)
    C.stream_gemm3(pw.wp, N, K, x, act, 2, nb, 1, K // (128 * nb), ss, 1e-5, pw.wscale, nw, wb, ft=2)
    xn = x.float() * torch.rsqrt((x.float() ** 2).mean(-1, keepdim=True) + 1e-5)
    ref_act = torch_ops.swiglu_interleaved(xn @ Wd.T).float()
    assert (act.float() - ref_act).abs().max().item() <= 2e-2 * ref_act.abs().max().item() + 2e-2
    # shapes the two-tile form is not built for are refused, not mis-launched
    with pytest.raises(RuntimeError):
        C.stream_gemm3(pw.wp, N, K, x, y, 0, nb, 1, K // (128 * nb), None, 0.0, pw.wscale, 7, wb, ft=2)


@pytest.mark.gpu
def test_mxfp4_expansion_is_the_packed_bf16_weight_bit_for_bit(C):
    """Wide steps run the bf16 tiled GEMM on an expansion of the MXFP4 matrix: the expansion equals,
    bit for bit, the dequantised weight packed as a bf16 weight would be (e2m1 x 2^e is exact in
    bf16), also when the scratch buffer is larger than the matrix; and the 254- and 2048-row
    shapes of a prompt step match the fp32 reference on the same quantised weights."""
    from kvmini.models.llama import dequantize_mxfp4, quantize_mxfp4
    from kvmini.ops import hip_ops

    torch.manual_seed(9)
    N, K = 1000, 1024  # N is padded to 1024 rows of zeros
    W = (torch.randn(N, K, device=_dev()) * 0.05).bfloat16()
    W[7] *= 51.0
    W[:, 96:128] *= 1e-4
    pw = hip_ops.PackedLinear(W, mxfp4=True)
    codes, scales = quantize_mxfp4(W)
    Wdq = dequantize_mxfp4(codes, scales, torch.bfloat16)
    Wdq = torch.cat((Wdq, torch.zeros(pw.n_pad - N, K, dtype=Wdq.dtype, device=Wdq.device)), 0)
    want = C.pack_weight(Wdq.contiguous()).view(-1).view(torch.int16)
    buf = torch.full((pw.n_pad * K + 4096,), 3.0, dtype=torch.bfloat16, device=_dev())
    C.unpack_mxfp4(pw.wp, pw.wscale, buf, pw.n_pad, K)
    assert torch.equal(buf[: pw.n_pad * K].view(torch.int16), want)
    assert (buf[pw.n_pad * K:] == 3.0).all()
    Wf = dequantize_mxfp4(codes, scales, torch.float32)
    for M in (254, 2048):
        x = torch.randn(M, K, device=_dev()).bfloat16()
        ref = x.float() @ Wf.T
        prev, hip_ops.FP8_ACTIVATIONS = hip_ops.FP8_ACTIVATIONS, False  # the expansion path (bf16 activations)
        try:
            y = hip_ops.linear(x, pw)
        finally:
            hip_ops.FP8_ACTIVATIONS = prev
        assert y.shape == (M, N) and (y.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item() + 1e-2
        ref8 = _fp8_rows(x) @ Wf.T  # the default: the tiled kernel reads the MXFP4 weights themselves
        y = hip_ops.linear(x, pw)
        assert y.shape == (M, N) and (y.float() - ref8).abs().max().item() <= 1e-2 * ref8.abs().max().item() + 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("M", [16, 48, 96, 200])
def test_stream_gemm_mxfp4_swiglu_and_folded_norm(M):
    from kvmini.models.llama import GU_GROUP, dequantize_mxfp4, quantize_mxfp4
    from kvmini.ops import hip_ops, torch_ops

    torch.manual_seed(6)
    K, I = 1024, 2048
    W = (torch.randn(2 * I, K, device=_dev()) * 0.04).bfloat16()
    x = torch.randn(M, K, device=_dev()).bfloat16()
    Wdq = dequantize_mxfp4(*quantize_mxfp4(W), torch.float32)
    pw = hip_ops.PackedLinear(W, mxfp4=True)
    xr = _fp8_rows(x) if M > 64 else x.float()  # steps of more than 64 rows: e4m3 activations on the scaled MFMA
    ref = torch_ops.swiglu_interleaved((xr @ Wdq.T))
    y = hip_ops.linear_swiglu(x, pw)
    assert y.shape == (M, I)
    assert (y.float() - ref.float()).abs().max().item() <= 2e-2 * ref.abs().max().item() + 2e-2
    # a Normed input: the row scale 1/rms is applied by the epilogue
    ss = (x.float() ** 2).view(M, -1, 512).sum(-1)
    yn = hip_ops.linear(hip_ops.Normed(x, ss.contiguous(), 1e-5), pw)
    refn = (xr * torch.rsqrt((x.float() ** 2).mean(-1, keepdim=True) + 1e-5)) @ Wdq.T
    assert (yn.float() - refn).abs().max().item() <= 2e-2 * refn.abs().max().item() + 2e-2
    assert GU_GROUP > 0


@pytest.mark.gpu
def test_engine_mxfp4_matches_fake_quantised_oracle():
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import fake_quantize_mxfp4, get_config, random_weights
    from kvmini.models.reference import dense_logits

    cfg = get_config("small-hd128")
    w = random_weights(cfg, "cuda:0", torch.bfloat16, seed=4)
    oracle = random_weights(cfg, "cuda:0", torch.bfloat16, seed=4)
    proj_bf16 = sum(t.numel() * 2 for lw in oracle.layers for t in (lw.wqkv, lw.wo, lw.wgu, lw.wdown))
    fake_quantize_mxfp4(oracle)
    for alias in ("mxfp4", "awq"):  # the reference's 4-bit profile names select the same format
        eng = LLMEngine(EngineConfig(model="small-hd128", device="cuda:0", max_num_seqs=8, max_batched_tokens=512, max_model_len=2048, num_blocks=256, kernels="hip", quantization=alias),
                        weights=w if alias == "mxfp4" else random_weights(cfg, "cuda:0", torch.bfloat16, seed=4))
        assert eng.weights.quantization == "mxfp4" and eng.weights.layers[0].wqkv.bits == 4
    eng.warmup()
    proj_fp4 = sum(t.wp.numel() + t.wscale.numel() * 4 for lw in eng.weights.layers for t in (lw.wqkv, lw.wo, lw.wgu, lw.wdown))
    assert proj_fp4 <= 0.27 * proj_bf16, (proj_fp4, proj_bf16)  # 4 bits + one scale byte per 32 weights = 0.266 of bf16
    prompts = [[1, 50, 60, 70, 80], list(range(1, 30)), [1] + [(7 * i) % 900 + 20 for i in range(299)]]
    outs = eng.generate(prompts, SamplingParams(max_tokens=8, ignore_eos=True))
    for p, o in zip(prompts, outs):
        ids = list(p)
        for tok in o:
            lg = dense_logits(oracle, ids)[-1]
            top = lg.max().item()
            assert lg[tok].item() >= top - 0.05 * abs(top) - 0.05, (p, o)
            ids.append(tok)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [16, 32, 64])
@pytest.mark.parametrize("N,K,fp8", [(4096, 4096, False), (4096, 14336, False), (2048, 8192, False), (4096, 4096, True), (256, 256, False)])
def test_linear_residual_epilogue_is_bit_equal_to_the_two_kernel_path(M, N, K, fp8, monkeypatch):
    """Split-K GEMM whose last workgroup folds the slabs into the residual == the same GEMM
    followed by residual_reduce: same slab order, same roundings, so bit-equal residual and (up to
    the grouping of the partial sums) equal sums of squares; repeated calls reuse the counters.
    (The engine leaves this epilogue off by default; the switch is turned on for the test.)"""
    from kvmini.ops import hip_ops

    monkeypatch.setattr(hip_ops, "FUSE_RESIDUAL", True)
    monkeypatch.setattr(hip_ops, "ROWSUM_GEMM", False)

    torch.manual_seed(12)
    W = (torch.randn(N, K, device=_dev()) * 0.03).bfloat16()
    pw = hip_ops.PackedLinear(W, fp8=fp8)
    for rep in range(3):
        x = torch.randn(M, K, device=_dev()).bfloat16()
        res0 = torch.randn(M, N, device=_dev()).bfloat16()
        want_res = res0.clone()
        part = hip_ops.linear(x, pw, allow_slabs=True)
        normed_ref, _ = hip_ops.add_rmsnorm(part, want_res, torch.ones(N, device=_dev(), dtype=torch.bfloat16), 1e-5)
        got_res = res0.clone()
        out = hip_ops.linear_residual(x, pw, got_res, 1e-5)
        assert out is not None
        normed, r2 = out
        assert r2 is got_res and normed.x is got_res
        if isinstance(part, hip_ops.Partial):  # same split plan on both sides: bit-equal
            assert torch.equal(got_res, want_res)
        else:
            assert (got_res.float() - want_res.float()).abs().max().item() <= 2e-2 * want_res.float().abs().max().item()
        ss_ref = (got_res.float() ** 2).sum(-1)
        assert torch.allclose(normed.ss.sum(-1), ss_ref, rtol=1e-4, atol=1e-3)
        assert torch.allclose(normed_ref.ss.sum(-1), normed.ss.sum(-1), rtol=1e-4, atol=1e-3)
    torch.cuda.synchronize()
    assert int(hip_ops._resid_counters[_dev()].abs().sum()) == 0
    assert hip_ops.linear_residual(torch.randn(20, K, device=_dev()).bfloat16(), pw, torch.zeros(20, N, device=_dev(), dtype=torch.bfloat16), 1e-5) is None  # 20 rows: not served


@pytest.mark.gpu
@pytest.mark.parametrize("M", [16, 32, 48, 64])
@pytest.mark.parametrize("N,K,fp8", [(4096, 4096, False), (4096, 14336, False), (2048, 8192, False), (4096, 4096, True), (1024, 2048, False), (8192, 28672, False)])
def test_rowsum_gemm_updates_the_residual_like_gemm_plus_residual_kernel(M, N, K, fp8, monkeypatch):
    """One workgroup per 16-feature tile, K split over its 8 waves, residual add in the epilogue:
    against the fp32 matmul of the same operands, and bit-equal to the split-K GEMM +
    residual_reduce pair whenever that pair splits K into the same 8 equal ranges (the waves' sums
    are added in slab order).  Sums of squares: one part per tile, totals equal the row's."""
    from kvmini.ops import hip_ops

    monkeypatch.setattr(hip_ops, "ROWSUM_GEMM", True)
    monkeypatch.setattr(hip_ops, "ROWSUM_MAX_ROWS_K", 1 << 30)
    torch.manual_seed(21)
    W = (torch.randn(N, K, device=_dev()) * 0.03).bfloat16()
    pw = hip_ops.PackedLinear(W, fp8=fp8)
    Wd = W.float()
    if fp8:
        from kvmini.models.llama import quantize_fp8_rows

        q, sc = quantize_fp8_rows(W)
        Wd = q.float() * sc[:, None]
    for nb in (2, 3, 4):
        if nb > 2 and M > 32:
            continue
        monkeypatch.setattr(hip_ops, "ROWSUM_NB", nb)
        x = torch.randn(M, K, device=_dev()).bfloat16()
        res0 = torch.randn(M, N, device=_dev()).bfloat16()
        got = res0.clone()
        out = hip_ops.linear_residual(x, pw, got, 1e-5)
        assert out is not None
        normed, r2 = out
        assert r2 is got and normed.x is got and normed.ss.shape == (M, N // 16)
        ref = (x.float() @ Wd.T).bfloat16().float() + res0.float()
        err = (got.float() - ref).abs().max().item()
        assert err <= 2e-2 * ref.abs().max().item(), (nb, err)
        assert torch.allclose(normed.ss.sum(-1), (got.float() ** 2).sum(-1), rtol=1e-4, atol=1e-3)
        assert torch.allclose(normed.ss, (got.float() ** 2).view(M, N // 16, 16).sum(-1), rtol=1e-4, atol=1e-4)
        part = hip_ops.linear(x, pw, allow_slabs=True)
        if isinstance(part, hip_ops.Partial) and part.slabs.shape[0] == 8 and hip_ops._plan_slabs(N, K)[2] * 256 * 8 == K:
            want = res0.clone()
            hip_ops.add_rmsnorm(part, want, torch.ones(N, device=_dev(), dtype=torch.bfloat16), 1e-5)
            assert torch.equal(got, want), nb
    assert hip_ops.linear_residual(torch.randn(20, K, device=_dev()).bfloat16(), pw, torch.zeros(20, N, device=_dev(), dtype=torch.bfloat16), 1e-5) is None


@pytest.mark.gpu
@pytest.mark.parametrize("parts", [8, 64, 256])
def test_projections_take_row_scales_from_any_number_of_partial_sums(parts):
    """The folded RMSNorm's 1/rms comes from per-row partial sums of squares — 8 from the residual
    kernel, hidden/16 from the row-sum GEMMs; the consuming GEMM epilogues give the same result for
    any split of the same total."""
    from kvmini.ops import hip_ops

    torch.manual_seed(5)
    M, N, K = 32, 1024, 4096
    W = (torch.randn(N, K, device=_dev()) * 0.03).bfloat16()
    pw = hip_ops.PackedLinear(W)
    x = torch.randn(M, K, device=_dev()).bfloat16()
    sq = (x.float() ** 2).view(M, parts, K // parts).sum(-1).contiguous()
    y = hip_ops.linear(hip_ops.Normed(x, sq, 1e-5), pw)
    ref = (x.float() * torch.rsqrt((x.float() ** 2).mean(-1, keepdim=True) + 1e-5)) @ W.float().T
    assert (y.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


def _decode_layers_pair(model, n_seqs, kv_dtype, graphs, max_tokens=12, prompt_len=9):
    """Two engines on the same random weights, one with the persistent decode-layers kernel, one with a
    kernel per phase; returns (tokens, KV cache, runner) of each after the same greedy generation."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    from kvmini.ops import hip_ops

    out = []
    # (the persistent kernel's attention phase is the dot-product form: the per-phase engine it is compared with bit
    # for bit keeps the fused decode kernel on that form too)
    hip_ops.native().set_attention_mfma_unsplit(False)
    for persistent in (True, False):
        hip_ops.DECODE_LAYERS = persistent  # read when the runner is built
        eng = LLMEngine(EngineConfig(model=model, device="cuda:0", max_num_seqs=max(n_seqs, 4), max_batched_tokens=128, max_model_len=1024, kernels="hip", num_blocks=512,
                                     kv_cache_dtype=kv_dtype, use_graphs=graphs, async_scheduling=False))
        if persistent:
            assert eng.runner.decode_layers is not None, "the persistent kernel does not apply to this model"
        else:
            eng.runner.decode_layers = None
        eng.warmup()
        prompts = [[1] + [(13 * i + 7 * j) % 900 + 20 for j in range(prompt_len + (i % 3))] for i in range(n_seqs)]
        toks = eng.generate(prompts, SamplingParams(max_tokens=max_tokens, ignore_eos=True))
        torch.cuda.synchronize()
        out.append((toks, eng.runner.kv_cache.clone(), eng.runner))
    hip_ops.DECODE_LAYERS = False
    hip_ops.native().set_attention_mfma_unsplit(True)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n_seqs,kv_dtype,graphs", [(3, "auto", False), (20, "auto", True), (40, "auto", True), (60, "auto", False), (20, "fp8", True)])
def test_persistent_decode_layers_equal_the_kernel_per_phase_path(n_seqs, kv_dtype, graphs):
    """All decoder layers in one persistent launch (phases separated by device-wide barriers): the
    GEMM phases use the streaming kernel's plans and slab order, so tokens AND the KV cache are
    bit-equal to the kernel-per-phase path — for 16 / 32 / 48 / 64-row steps, eagerly and through
    HIP graphs, with a bf16 and an FP8 cache.  No barrier may have timed out."""
    (tok_a, kv_a, run_a), (tok_b, kv_b, _run_b) = _decode_layers_pair("mid-hd128", n_seqs, kv_dtype, graphs)
    assert run_a.decode_layers.error_word() == 0
    assert tok_a == tok_b
    assert kv_a.shape == kv_b.shape
    diff = (kv_a.view(torch.uint8) != kv_b.view(torch.uint8))
    assert not bool(diff.any()), f"{int(diff.sum())} KV bytes differ, first at {diff.nonzero()[0].tolist()}"


@pytest.mark.gpu
def test_token_gap_kernels_equal_the_per_event_model_and_feed_the_histogram():
    """Raw (request, index, t_ns) events -> gaps on the device (hash insert / emit / commit), over
    batches cut anywhere; the gaps binned by stream_hist_kernel give the quantiles of the model's gaps."""
    import numpy as np

    from kvmini.ops import hist_quantile as hq
    from test_loadgen import _event_batches, _gaps_model

    batches = _event_batches(5, n_req=2000, n_batches=9)
    want = _gaps_model(batches)
    tg = hq.TokenGaps(device="cuda:0", requests_hint=16)
    hist = hq.StreamingHistogram(1, device="cuda:0")
    allg = []
    for ev, w in zip(batches, want):
        g = tg.gaps(ev)
        assert g.shape[0] == ev.shape[0] and g.dtype == torch.float32
        got = g[~torch.isnan(g)].sort().values.cpu().numpy()
        assert got.shape[0] == len(w) and np.allclose(got, np.asarray(w, dtype=np.float32), rtol=1e-6), (got.shape, len(w))
        hist.update(g)
        allg += w
    st = hist.snapshot((0.5, 0.95, 0.99))[0]
    assert int(st["count"]) == len(allg)
    ref = np.quantile(np.asarray(allg), (0.5, 0.95, 0.99))
    for q, r in zip((0.5, 0.95, 0.99), ref):
        assert abs(st["quantiles"][q] - r) <= 0.02 * r


# ---------------------------------------------------------------- block-scaled MFMA (FP8 x FP8, FP4)
def _e4m3_bytes(vals):
    return vals.to(torch.float8_e4m3fn).view(torch.uint8)


_E2M1 = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)


@pytest.mark.gpu
def test_scaled_mfma_operand_map_with_exact_integer_data(C):
    """One v_mfma_scale_f32_16x16x128_f8f6f4 on hand-built registers pins the operand map the FP8
    GEMM relies on (found with one-hot probes, ``scripts/mfma_scale_map.py``).  Lane l = (r = l & 15,
    q = l >> 4) supplies row r of A / column r of B:
      * 8-bit formats: bytes 0..15 of its eight dwords are k = 16 q + j, bytes 16..31 are k = 64 + 16 q + j;
      * FP4: nibble j (low nibble first) of its first four dwords is k = 32 q + j;
      * the scale byte of lane (r, b) scales the 32 elements k = 32 b .. 32 b + 31 of row / column r;
      * D in the 16x16 C layout: lane (c, q), register e = D[4 q + e][c].
    Asymmetric integer data and power-of-two scales: the comparison is exact."""
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-4, 5, (16, 128), generator=g).float()
    B = torch.randint(-3, 4, (128, 16), generator=g).float()
    A[:, 5] += torch.arange(16).float() % 3  # rows differ
    ea = torch.randint(-2, 3, (16, 4), generator=g)  # block exponents per (row, 32-block)
    eb = torch.randint(-2, 3, (16, 4), generator=g)  # per (column, 32-block)
    lanes = torch.arange(64)
    r, q = lanes & 15, lanes >> 4
    k8 = torch.stack([torch.cat((torch.arange(16) + 16 * qq, torch.arange(16) + 64 + 16 * qq)) for qq in range(4)])  # [q, 32] -> k
    a_reg = _e4m3_bytes(torch.stack([A[r[l], k8[q[l]]] for l in range(64)])).contiguous().view(torch.int32).view(64, 8)
    b_reg = _e4m3_bytes(torch.stack([B[k8[q[l]], r[l]] for l in range(64)])).contiguous().view(torch.int32).view(64, 8)
    sa = (127 + ea[r, q]).int()
    sb = (127 + eb[r, q]).int()
    out = C.mfma_scale_probe(a_reg.to(_dev()), b_reg.to(_dev()), sa.to(_dev()), sb.to(_dev()), 0, 0).cpu()
    As = A * (2.0 ** ea.float()).repeat_interleave(32, dim=1)
    Bs = B * (2.0 ** eb.float()).repeat_interleave(32, dim=1).T
    D = As @ Bs
    want = torch.stack([D[4 * q[l] : 4 * q[l] + 4, r[l]] for l in range(64)])
    assert torch.equal(out, want), (out - want).abs().max()
    # FP4 (e2m1) A against FP8 B: 32 nibbles in the first four dwords, k = 32 q + j
    codes = torch.randint(0, 16, (16, 128), generator=g)
    A4 = torch.tensor(_E2M1)[codes & 7] * torch.where(codes >= 8, -1.0, 1.0)
    nib = torch.stack([codes[r[l], 32 * q[l] : 32 * q[l] + 32] for l in range(64)])  # [64, 32]
    packed = (nib[:, 0::2] | (nib[:, 1::2] << 4)).to(torch.uint8)  # [64, 16] bytes
    a4_reg = torch.cat([packed, torch.zeros(64, 16, dtype=torch.uint8)], 1).contiguous().view(torch.int32).view(64, 8)
    out4 = C.mfma_scale_probe(a4_reg.to(_dev()), b_reg.to(_dev()), sa.to(_dev()), sb.to(_dev()), 4, 0).cpu()
    D4 = (A4 * (2.0 ** ea.float()).repeat_interleave(32, dim=1)) @ Bs
    want4 = torch.stack([D4[4 * q[l] : 4 * q[l] + 4, r[l]] for l in range(64)])
    assert torch.equal(out4, want4), (out4 - want4).abs().max()


@pytest.mark.gpu
def test_quantize_rows_fp8_scale_and_fragment_order(C):
    torch.manual_seed(2)
    M, K = 37, 512
    x = (torch.randn(M, K + 64, device=_dev()) * 3).bfloat16()[:, :K]  # strided rows
    x[3] = 0
    x8, sc = C.quantize_rows_fp8(x)
    amax = x.float().abs().amax(1)
    want_sc = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.allclose(sc, want_sc, rtol=1e-6)
    q = (x.float() / want_sc[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)  # [M, K] in natural k order
    # inside every 64 elements: byte 16 qq + 8 half + j <- k = 32 half + 8 qq + j
    nat = q.view(M, K // 64, 2, 4, 8)  # group, half, qq, j
    perm = nat.permute(0, 1, 3, 2, 4).contiguous().view(M, K)
    assert torch.equal(x8, perm)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [129, 254, 512, 1024, 2048])
@pytest.mark.parametrize("N,K,epi", [(4096, 4096, 0), (6144, 4096, 0), (28672, 4096, 2), (4096, 14336, 0)])
def test_fp8_weights_times_fp8_activations_matches_fp32_matmul_of_the_quantised_operands(M, N, K, epi):
    """The wide step of an FP8 model: activations quantised per token to e4m3, weights per row,
    multiplied on the scaled MFMA.  Reference: fp32 matmul of the DEQUANTISED operands (what the
    instruction computes exactly, up to fp32 summation order), bf16 rounding of the output on top."""
    from kvmini.models.llama import interleave_gate_up
    from kvmini.ops import hip_ops

    torch.manual_seed(M + N)
    W = torch.randn(N, K, device=_dev()) * 0.03
    if epi == 2:
        W = interleave_gate_up(W[: N // 2], W[N // 2 :]).contiguous()
    pw = hip_ops.PackedLinear(W.bfloat16(), fp8=True)
    x = (torch.randn(M, K, device=_dev()) * 1.5).bfloat16()
    x[:, 7] *= 20  # an outlier column: the per-token scale follows it
    assert hip_ops.FP8_ACTIVATIONS
    y = hip_ops._wide(pw, x, epi, allow_slabs=False)
    # reference from the quantised operands
    C = hip_ops.native()
    _x8, sc = C.quantize_rows_fp8(x)
    xq = (x.float() / sc[:, None]).to(torch.float8_e4m3fn).float() * sc[:, None]
    from kvmini.models.llama import quantize_fp8_rows

    qw, wsc = quantize_fp8_rows(W.bfloat16())
    wq = qw.float() * wsc.float().view(-1, 1)
    ref = xq @ wq.T
    if epi == 2:
        half = ref.view(M, N // 16, 2, 8)
        ref = (torch.nn.functional.silu(half[:, :, 0]) * half[:, :, 1]).reshape(M, N // 2)
    err = (y.float() - ref).abs().max().item()
    tol = 1.2e-2 * ref.abs().max().item() + 1e-2
    assert y.shape == ref.shape and err <= tol, (err, tol)
    # and it stays close to the unquantised product (what the model wanted)
    full = x.float() @ W.bfloat16().float().T
    if epi == 0:
        rel = (y.float() - full).norm() / full.norm()
        assert rel < 0.06, rel


@pytest.mark.gpu
@pytest.mark.parametrize("M", [129, 254, 512, 1024, 2048])
@pytest.mark.parametrize("N,K,epi", [(4096, 4096, 0), (6144, 4096, 0), (28672, 4096, 2), (4096, 14336, 0)])
def test_mxfp4_weights_times_fp8_activations_matches_fp32_matmul_of_the_quantised_operands(M, N, K, epi):
    """The wide step of an MXFP4 model on the scaled MFMA: the packed e2m1 fragments are transposed
    across the lane quarters into the FP4 operand, their E8M0 block scales are applied by the
    instruction, the activations are per-token e4m3.  Reference: fp32 matmul of the dequantised
    operands."""
    from kvmini.models.llama import dequantize_mxfp4, interleave_gate_up, quantize_mxfp4
    from kvmini.ops import hip_ops

    torch.manual_seed(M + N + 1)
    W = torch.randn(N, K, device=_dev()) * 0.03
    W[:, 100:132] *= 8  # blocks with very different scales
    if epi == 2:
        W = interleave_gate_up(W[: N // 2], W[N // 2 :]).contiguous()
    Wb = W.bfloat16()
    pw = hip_ops.PackedLinear(Wb.clone(), mxfp4=True)
    x = (torch.randn(M, K, device=_dev()) * 1.5).bfloat16()
    x[:, 7] *= 20
    y = hip_ops._stream_fp4(pw, x, epi, False)
    codes, scales = quantize_mxfp4(Wb)
    wq = dequantize_mxfp4(codes, scales, torch.float32)
    C = hip_ops.native()
    _x8, sc = C.quantize_rows_fp8(x, False)
    xq = (x.float() / sc[:, None]).to(torch.float8_e4m3fn).float() * sc[:, None]
    ref = xq @ wq.T
    if epi == 2:
        half = ref.view(M, N // 16, 2, 8)
        ref = (torch.nn.functional.silu(half[:, :, 0]) * half[:, :, 1]).reshape(M, N // 2)
    err = (y.float() - ref).abs().max().item()
    tol = 1.2e-2 * ref.abs().max().item() + 1e-2
    assert y.shape == ref.shape and err <= tol, (err, tol)
