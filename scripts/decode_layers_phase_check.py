#!/usr/bin/env python3
"""Phase-by-phase check of the persistent decode-layers kernel on one layer: run it on a crafted
residual stream with every row a live one-token sequence, then compare each workspace region with
the result of the corresponding stand-alone kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KVMINI_DECODE_LAYERS", "1")
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.ops import hip_ops, native

model = sys.argv[1] if len(sys.argv) > 1 else "mid-hd128"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
eng = LLMEngine(EngineConfig(model=model, device="cuda:0", max_num_seqs=64, max_batched_tokens=128, max_model_len=1024, kernels="hip", kv_memory_fraction=0.05, use_graphs=False))
r = eng.runner
dl = r.decode_layers
C = native()
dev = r.device
H = r.cfg.hidden_size
lay = C.decode_layers_layout(H, r.nh, r.nkv, dl.inter, dl.s_qkv, dl.s_proj)
names = ["flags", "state", "xres", "ssb", "qkv_slabs", "attn_out", "proj_slabs", "act", "total"]
off = dict(zip(names, lay))
print("plans", dl.plans, "layout", off)
torch.manual_seed(3)
residual = torch.randn(T, H, device=dev).bfloat16()
parts = (H + 511) // 512
ss = (residual.float() ** 2).view(T, parts, -1).sum(-1).contiguous()
# step inputs: row t = sequence t at position 5 (ctx 6), slot from block t
positions = torch.full((T,), 5, dtype=torch.int32, device=dev)
tok_seq = torch.arange(T, dtype=torch.int32, device=dev)
ctx = torch.full((T,), 6, dtype=torch.int32, device=dev)
bt = torch.zeros((64, r.max_blocks_per_seq), dtype=torch.int32, device=dev)
bt[:T, 0] = torch.arange(1, T + 1, dtype=torch.int32, device=dev)
slots = (bt[:T, 0] * 16 + 5).to(torch.int32)
r.kv_cache.copy_(torch.randn(r.kv_cache.shape, device=dev).to(r.kv_cache.dtype))
kv_before = r.kv_cache.clone()

# one-layer launch
one = hip_ops.DecodeLayers.__new__(hip_ops.DecodeLayers)
one.__dict__.update(dl.__dict__)
one.table = dl.table[:1].contiguous()
res_a, ss_a = residual.clone(), ss.clone()
one.run(res_a, ss_a, positions, r.cos, r.sin, slots, bt, tok_seq, ctx, r.cfg.rms_eps, r.scale)
torch.cuda.synchronize()
print("error word", dl.error_word())
ws = dl.ws
def region(name, dtype, shape):
    n = 1
    for s_ in shape:
        n *= s_
    nbytes = n * torch.tensor([], dtype=dtype).element_size()
    return ws[off[name] : off[name] + nbytes].view(dtype).view(*shape).clone()
kv_a = r.kv_cache.clone()

# reference, phase by phase
r.kv_cache.copy_(kv_before)
lw = r.w.layers[0]
h = hip_ops.Normed(residual.clone(), ss.clone(), r.cfg.rms_eps)
qkv = hip_ops.linear(h, lw.wqkv, allow_slabs=True)
print("qkv ref type", type(qkv).__name__, getattr(qkv, "slabs", qkv).shape)
S = dl.plans[3]
qw = (r.nh + 2 * r.nkv) * 128
got_slabs = region("qkv_slabs", torch.float32, (S, T, qw))
if isinstance(qkv, hip_ops.Partial):
    d = (got_slabs - qkv.slabs).abs()
    print("qkv slabs equal:", torch.equal(got_slabs, qkv.slabs), "max|d|", d.max().item(), "ref max", qkv.slabs.abs().max().item())
    for s_ in range(S):
        print("  slab", s_, "max|d|", d[s_].max().item(), "rows bad", (d[s_].amax(-1) > 0).nonzero().flatten().tolist()[:20])
a = hip_ops.rope_attention_fused(qkv, positions, r.cos, r.sin, r.kv_cache[0, 0], r.kv_cache[0, 1], slots, bt, tok_seq, ctx, r.nh, r.scale, n_split=1)
got_attn = region("attn_out", torch.bfloat16, (T, r.nh * 128))
print("attn equal:", torch.equal(got_attn, a), "max|d|", (got_attn.float() - a.float()).abs().max().item())
print("layer-0 KV equal:", torch.equal(kv_a[0], r.kv_cache[0]))
hh, res_ref = hip_ops.add_rmsnorm(hip_ops.linear(a, lw.wo, allow_slabs=True), residual.clone(), lw.mlp_norm, r.cfg.rms_eps)
act = hip_ops.linear_swiglu(hh, lw.wgu)
got_act = region("act", torch.bfloat16, (T, dl.inter))
print("act equal:", torch.equal(got_act, act), "max|d|", (got_act.float() - act.float()).abs().max().item(), "ref max", act.float().abs().max().item())
h2, res2 = hip_ops.add_rmsnorm(hip_ops.linear(act, lw.wdown, allow_slabs=True), res_ref, lw.attn_norm, r.cfg.rms_eps)
print("final residual equal:", torch.equal(res_a, res2), "max|d|", (res_a.float() - res2.float()).abs().max().item())
print("final ss close:", torch.allclose(ss_a, h2.ss), (ss_a - h2.ss).abs().max().item())

if isinstance(qkv, hip_ops.Partial):
    g, rf = got_slabs, qkv.slabs
    print("got[0,0,:8]", g[0, 0, :8].tolist())
    print("ref[0,0,:8]", rf[0, 0, :8].tolist())
    # does a got slab equal another ref slab?
    for a_ in range(S):
        for b_ in range(S):
            if torch.allclose(g[a_], rf[b_], atol=1e-3):
                print("got slab", a_, "== ref slab", b_)
    tot_g, tot_r = g.sum(0), rf.sum(0)
    print("sum over slabs equal-ish:", torch.allclose(tot_g, tot_r, atol=1e-2), (tot_g - tot_r).abs().max().item())
    # per 16-column tile: which ref tile does got tile 0 match
    for tg in range(4):
        best = min(range(qw // 16), key=lambda t_: (g[0, :, tg * 16 : tg * 16 + 16] - rf[0, :, t_ * 16 : t_ * 16 + 16]).abs().max().item())
        print("got tile", tg, "closest ref tile", best, (g[0, :, tg * 16 : tg * 16 + 16] - rf[0, :, best * 16 : best * 16 + 16]).abs().max().item())
    print("zero fraction of got", (g == 0).float().mean().item())
