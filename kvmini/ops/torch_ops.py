"""Plain-PyTorch implementations of the engine's ops.

These are the numerics oracles for the HIP kernels (tests compare each kernel
against these in fp32) and the execution path on CPU.  Shapes:

* token-major activations ``[T, hidden]`` — decode tokens and prefill-chunk
  tokens of many sequences are flattened into one ragged batch;
* KV cache per layer ``[num_blocks, n_kv_heads, block_size, head_dim]`` (K and V
  separately) so one (block, kv-head) tile is contiguous;
* ``slot_mapping[t]`` = ``block * block_size + offset`` where token ``t``'s K/V go
  (-1 = padding token, nothing written);
* ``tok_seq[t]`` = row of ``block_tables`` for token ``t``; ``ctx_lens[t]`` = number of
  cache positions token ``t`` attends to (its own position + 1; 0 = padding).
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from kvmini.models.llama import GU_GROUP


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return (y * w.float()).to(x.dtype)


def add_rmsnorm(x: torch.Tensor, residual: "torch.Tensor | None", w: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """residual' = residual + x (kept in the activation dtype; ``residual=None`` means
    zero, i.e. the first layer); y = rmsnorm(residual')."""
    r = x if residual is None else (residual.float() + x.float()).to(x.dtype)
    return rmsnorm(r, w, eps), r


def embed_tokens(table: torch.Tensor, tokens: torch.Tensor, feed: torch.Tensor, prev_ids: torch.Tensor) -> torch.Tensor:
    """Embedding rows of a step's tokens.  ``feed[i] = r + 1`` takes the id from
    ``prev_ids[r]`` (the previous step's sampled ids, still on the device) instead of
    ``tokens[i]``; 0 = use ``tokens[i]``.  Ids outside the vocabulary read row 0."""
    fed = prev_ids[(feed.long() - 1).clamp(0, prev_ids.numel() - 1)]
    tok = torch.where((feed > 0) & (feed <= prev_ids.numel()), fed, tokens.long())
    tok = torch.where((tok < 0) | (tok >= table.shape[0]), torch.zeros_like(tok), tok)
    return table[tok]


def select_rows(h: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    return torch.index_select(h, 0, idx.long() if h.device.type == "cpu" else idx)


def linear(x: torch.Tensor, w: torch.Tensor, allow_slabs: bool = False) -> torch.Tensor:
    """``allow_slabs`` is a hint for the HIP path (split-K partial sums); ignored here."""
    return F.linear(x, w)


def add_bias(y: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``y + b`` for a projection's output ``[T, N]`` and its bias ``[N]``."""
    return y + b.to(y.dtype)


def rope_kv_write(
    qkv: torch.Tensor,
    positions: torch.Tensor,
    cos: torch.Tensor,
    sin: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    slot_mapping: torch.Tensor,
    n_heads: int,
    n_kv: int,
    head_dim: int,
    kv_scale: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """Split fused qkv ``[T, (nh + 2 nkv) * hd]``, rotate q and k (rotate-half
    pairing), scatter k/v into the paged cache, return rotated q ``[T, nh, hd]``.
    ``kv_scale`` (fp32 ``[2, nkv]``): FP8 cache — e4m3 of ``x / scale``, clamped to +-448."""
    t = qkv.shape[0]
    q, k, v = qkv.split((n_heads * head_dim, n_kv * head_dim, n_kv * head_dim), dim=-1)
    q = q.view(t, n_heads, head_dim)
    k = k.view(t, n_kv, head_dim)
    v = v.view(t, n_kv, head_dim)
    c = cos[positions][:, None, :]
    s = sin[positions][:, None, :]
    half = head_dim // 2

    def rot(x: torch.Tensor) -> torch.Tensor:
        xf = x.float()
        x1, x2 = xf[..., :half], xf[..., half:]
        return torch.cat((x1 * c - x2 * s, x2 * c + x1 * s), dim=-1).to(x.dtype)

    q = rot(q)
    k = rot(k)
    block_size = k_cache.shape[2]
    valid = slot_mapping >= 0
    slots = slot_mapping.clamp_min(0)
    blk = slots // block_size
    off = slots % block_size
    if kv_scale is not None:
        # float8 tensors have no indexed assignment on every backend: move the bytes
        k = (k.float() / kv_scale[0][None, :, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        v = (v.float() / kv_scale[1][None, :, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        k_cache, v_cache = k_cache.view(torch.uint8), v_cache.view(torch.uint8)
    if bool(valid.all()):
        k_cache[blk, :, off] = k
        v_cache[blk, :, off] = v
    else:
        k_cache[blk[valid], :, off[valid]] = k[valid]
        v_cache[blk[valid], :, off[valid]] = v[valid]
    return q


def paged_attention(
    q: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    block_tables: torch.Tensor,
    tok_seq: torch.Tensor,
    ctx_lens: torch.Tensor,
    scale: float,
    max_blocks: int | None = None,
    kv_scale: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """Attention of every token over its sequence's cached K/V (GQA aware).
    Returns ``[T, n_heads * head_dim]``; padding tokens (ctx_len 0) give zeros."""
    t, nh, hd = q.shape
    nkv, bs = k_cache.shape[1], k_cache.shape[2]
    rep = nh // nkv
    bt = block_tables[tok_seq.long()]
    if max_blocks is not None:
        bt = bt[:, :max_blocks]
    mb = bt.shape[1]
    if kv_scale is not None:
        k = k_cache.view(torch.uint8)[bt.long()].view(torch.float8_e4m3fn).permute(0, 2, 1, 3, 4).reshape(t, nkv, mb * bs, hd).float() * kv_scale[0][None, :, None, None]
        v = v_cache.view(torch.uint8)[bt.long()].view(torch.float8_e4m3fn).permute(0, 2, 1, 3, 4).reshape(t, nkv, mb * bs, hd).float() * kv_scale[1][None, :, None, None]
    else:
        k = k_cache[bt.long()].permute(0, 2, 1, 3, 4).reshape(t, nkv, mb * bs, hd).float()
        v = v_cache[bt.long()].permute(0, 2, 1, 3, 4).reshape(t, nkv, mb * bs, hd).float()
    qf = q.view(t, nkv, rep, hd).float()
    scores = torch.einsum("tgrd,tgsd->tgrs", qf, k) * scale
    mask = torch.arange(mb * bs, device=q.device)[None, :] < ctx_lens[:, None]
    scores = scores.masked_fill(~mask[:, None, None, :], float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    probs = torch.nan_to_num(probs, nan=0.0)  # all-masked rows (padding)
    out = torch.einsum("tgrs,tgsd->tgrd", probs, v)
    return out.reshape(t, nh * hd).to(q.dtype)


def swiglu_interleaved(gu: torch.Tensor) -> torch.Tensor:
    """``silu(gate) * up`` for the interleaved gate/up layout (groups of ``GU_GROUP`` rows)."""
    *lead, n = gu.shape
    x = gu.reshape(*lead, n // (2 * GU_GROUP), 2, GU_GROUP).float()
    return (F.silu(x[..., 0, :]) * x[..., 1, :]).reshape(*lead, n // 2).to(gu.dtype)


def linear_swiglu(x: torch.Tensor, wgu: torch.Tensor) -> torch.Tensor:
    return swiglu_interleaved(F.linear(x, wgu))


def greedy_sample(logits: torch.Tensor) -> torch.Tensor:
    return logits.argmax(dim=-1)


def guided_mask(logits: torch.Tensor, rows: torch.Tensor, slots: torch.Tensor, init: torch.Tensor, state: torch.Tensor, cls: torch.Tensor, vc: int,
                script: Optional[torch.Tensor] = None, cursor: Optional[torch.Tensor] = None, cur_init: Optional[torch.Tensor] = None) -> None:
    """Reference of the guided-decoding mask: the Python automaton, row by row."""
    from kvmini.engine.guided import mask_rows_reference

    mask_rows_reference(logits, rows.tolist(), slots.tolist(), init.view(-1, 4).tolist(), state, cls.cpu().numpy(), vc, script, cursor, None if cur_init is None else cur_init.tolist())


def guided_advance(out_ids: torch.Tensor, rows: torch.Tensor, slots: torch.Tensor, state: torch.Tensor, cls: torch.Tensor, script: Optional[torch.Tensor] = None,
                   cursor: Optional[torch.Tensor] = None) -> None:
    from kvmini.engine.guided import advance_rows_reference

    advance_rows_reference(out_ids, rows.tolist(), slots.tolist(), state, cls.cpu().numpy(), script, cursor)
