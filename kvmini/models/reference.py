"""Dense, cache-free fp32 reference forward of the Llama decoder.

Used only as a test oracle: it recomputes the whole sequence with ordinary
causal attention, so the engine's paged, ragged, chunked, graph-replayed path
can be checked token-for-token against it.
"""

from __future__ import annotations

from typing import List

import torch
import torch.nn.functional as F

from .llama import LlamaWeights, deinterleave_gate_up, rope_tables


@torch.no_grad()
def dense_logits(w: LlamaWeights, token_ids: List[int]) -> torch.Tensor:
    """Logits [T, vocab] (fp32) for one sequence.  TP-sharded weights are not
    supported here (build the oracle from tp_size=1 weights)."""
    assert w.tp_size == 1
    cfg = w.cfg
    dev = w.device
    t = len(token_ids)
    ids = torch.tensor(token_ids, device=dev, dtype=torch.long)
    cos, sin = rope_tables(cfg, t, dev)
    x = w.embed[ids].float()
    nh, nkv, hd = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
    half = hd // 2

    def norm(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
        return v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + cfg.rms_eps) * g.float()

    def rot(v: torch.Tensor) -> torch.Tensor:
        v1, v2 = v[..., :half], v[..., half:]
        c, s = cos[:, None, :], sin[:, None, :]
        return torch.cat((v1 * c - v2 * s, v2 * c + v1 * s), -1)

    mask = torch.ones(t, t, dtype=torch.bool, device=dev).tril()
    for lw in w.layers:
        h = norm(x, lw.attn_norm)
        qkv = h @ lw.wqkv.float().T
        if lw.bqkv is not None:
            qkv = qkv + lw.bqkv.float()
        q, k, v = qkv.split((nh * hd, nkv * hd, nkv * hd), -1)
        q = rot(q.view(t, nh, hd))
        k = rot(k.view(t, nkv, hd))
        v = v.view(t, nkv, hd)
        rep = nh // nkv
        k = k.repeat_interleave(rep, 1)
        v = v.repeat_interleave(rep, 1)
        sc = torch.einsum("qhd,khd->hqk", q, k) * cfg.softmax_scale
        sc = sc.masked_fill(~mask[None], float("-inf"))
        a = torch.einsum("hqk,khd->qhd", sc.softmax(-1), v).reshape(t, nh * hd)
        x = x + a @ lw.wo.float().T
        h = norm(x, lw.mlp_norm)
        g, u = deinterleave_gate_up(h @ lw.wgu.float().T)
        x = x + (F.silu(g) * u) @ lw.wdown.float().T
    return norm(x, w.final_norm) @ w.lm_head.float().T


@torch.no_grad()
def dense_greedy(w: LlamaWeights, prompt: List[int], n: int) -> List[int]:
    ids = list(prompt)
    out = []
    for _ in range(n):
        tok = int(dense_logits(w, ids)[-1].argmax())
        out.append(tok)
        ids.append(tok)
    return out
