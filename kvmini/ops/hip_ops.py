"""Engine ops backed by the gfx950 HIP kernels (same signatures as ``torch_ops``).

Linear layers run on the weight-streaming MFMA kernel (``csrc/stream_gemm.hip``)
with weights re-ordered once, at load time, into MFMA fragment order
(:class:`PackedLinear`).  For narrow outputs the kernel splits K across
workgroups and returns fp32 partial-sum *slabs* ``[splits, T, N]``; the next op
(:func:`add_rmsnorm` or :func:`rope_kv_write`) accepts slabs and sums them in its
prologue, so the reduction costs no extra launch.  Batches above 64 token rows
are processed in 64-row pieces (the weights are re-streamed per piece).
"""

from __future__ import annotations

from typing import Any, Optional, Tuple, Union

import torch
import torch.nn.functional as F

from . import gpu_available, native

MAX_M = 64
KS_PER_CHUNK = 8  # k-steps (of 32) per pipelined chunk: 8 KiB of weights in flight per wave
MAX_SPLITS = 16


def require() -> None:
    """Fail loudly when the HIP path cannot run (missing build or no GPU)."""
    native()
    if not gpu_available():
        raise RuntimeError("kvmini HIP kernels need a ROCm GPU (torch.cuda.is_available() is False)")


class PackedLinear:
    """A bf16 ``[N, K]`` weight re-ordered into MFMA fragment order
    ``[N/16][K/32][64][8]`` (N zero-padded to a multiple of 64)."""

    __slots__ = ("wp", "n", "k", "n_pad", "ks")

    def __init__(self, w: torch.Tensor):
        n, k = w.shape
        if w.dtype != torch.bfloat16 or not w.is_cuda:
            raise ValueError("PackedLinear needs a bf16 GPU weight")
        ks = KS_PER_CHUNK if k % (32 * KS_PER_CHUNK) == 0 else 4
        if k % (32 * ks):
            raise ValueError(f"in_features {k} must be a multiple of 128 for the streaming GEMM")
        n_pad = (n + 63) // 64 * 64
        if n_pad != n:
            w = torch.cat((w, torch.zeros(n_pad - n, k, dtype=w.dtype, device=w.device)), 0)
        self.wp = native().pack_weight(w.contiguous())
        self.n, self.k, self.n_pad, self.ks = n, k, n_pad, ks

    @property
    def shape(self) -> Tuple[int, int]:
        return (self.n, self.k)

    def numel(self) -> int:
        return self.wp.numel()

    def element_size(self) -> int:
        return self.wp.element_size()


Weight = Union[torch.Tensor, PackedLinear]


def prepare_weights(weights: Any) -> None:
    """Pack every linear weight of a ``LlamaWeights`` in place (idempotent)."""
    def pk(w: Weight) -> Weight:
        return w if isinstance(w, PackedLinear) else PackedLinear(w)

    for lw in weights.layers:
        lw.wqkv, lw.wo, lw.wgu, lw.wdown = pk(lw.wqkv), pk(lw.wo), pk(lw.wgu), pk(lw.wdown)
    if weights.lm_head is weights.embed:
        weights.lm_head = PackedLinear(weights.embed.clone())
    else:
        weights.lm_head = pk(weights.lm_head)
    torch.cuda.empty_cache()


def _pad_rows(x: torch.Tensor) -> Tuple[torch.Tensor, int]:
    m = x.shape[0]
    mp = (m + 15) // 16 * 16
    if mp == m:
        return x, m
    xp = torch.zeros((mp, x.shape[1]), dtype=x.dtype, device=x.device)
    xp[:m] = x
    return xp, m


def _stream(w: PackedLinear, x: torch.Tensor, epi: int, allow_slabs: bool) -> torch.Tensor:
    C = native()
    x, m = _pad_rows(x)
    mp = x.shape[0]
    n_out = w.n_pad // 2 if epi == 2 else w.n_pad
    if mp <= MAX_M:
        if allow_slabs and epi == 0:
            splits, cps = C.stream_gemm_plan(w.n_pad, w.k, w.ks, MAX_SPLITS)
            if splits > 1:
                out = torch.empty((splits, mp, w.n_pad), dtype=torch.float32, device=x.device)
                C.stream_gemm(w.wp, w.n_pad, w.k, x, out, 1, w.ks, splits, cps)
                return out[:, :m] if m != mp else out
        nchunks = (w.k // 32) // w.ks
        out = torch.empty((mp, n_out), dtype=torch.bfloat16, device=x.device)
        C.stream_gemm(w.wp, w.n_pad, w.k, x, out, epi, w.ks, 1, nchunks)
    else:
        nchunks = (w.k // 32) // w.ks
        out = torch.empty((mp, n_out), dtype=torch.bfloat16, device=x.device)
        for r0 in range(0, mp, MAX_M):
            r1 = min(mp, r0 + MAX_M)
            C.stream_gemm(w.wp, w.n_pad, w.k, x[r0:r1], out[r0:r1], epi, w.ks, 1, nchunks)
    n_true = w.n // 2 if epi == 2 else w.n
    if m != mp or n_true != n_out:
        out = out[:m, :n_true]
    return out


def add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, w: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """``x``: bf16 ``[T, H]`` or fp32 slabs ``[S, T, H]``; ``residual`` is updated in place."""
    y = torch.empty_like(residual)
    if x.dim() == 3:
        native().add_rmsnorm_slabs(x.contiguous(), residual, w, y, eps)
    else:
        native().add_rmsnorm(x, residual, w, y, eps)
    return y, residual


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    y, _ = add_rmsnorm(x, torch.zeros_like(x), w, eps)
    return y


def linear(x: torch.Tensor, w: Weight, allow_slabs: bool = False) -> torch.Tensor:
    if isinstance(w, PackedLinear):
        return _stream(w, x, 0, allow_slabs)
    return F.linear(x, w)


def linear_swiglu(x: torch.Tensor, wgu: Weight) -> torch.Tensor:
    if isinstance(wgu, PackedLinear):
        return _stream(wgu, x, 2, False)
    gu = F.linear(x, wgu)
    act = torch.empty((gu.shape[0], gu.shape[1] // 2), dtype=gu.dtype, device=gu.device)
    native().swiglu_interleaved(gu, act)
    return act


def swiglu_interleaved(gu: torch.Tensor) -> torch.Tensor:
    act = torch.empty((gu.shape[0], gu.shape[1] // 2), dtype=gu.dtype, device=gu.device)
    native().swiglu_interleaved(gu, act)
    return act


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
) -> torch.Tensor:
    """``qkv``: bf16 ``[T, W]`` (q is rotated in place and returned as a strided
    view) or fp32 slabs ``[S, T, W]`` (q is written to a new ``[T, nh, hd]``)."""
    if qkv.dim() == 3:
        t = qkv.shape[1]
        q = torch.empty((t, n_heads, head_dim), dtype=torch.bfloat16, device=qkv.device)
        native().rope_kv_write_slabs(qkv.contiguous(), q, positions, cos, sin, k_cache, v_cache, slot_mapping, n_heads, n_kv, head_dim)
        return q
    if not qkv.is_contiguous():
        qkv = qkv.contiguous()
    native().rope_kv_write(qkv, positions, cos, sin, k_cache, v_cache, slot_mapping, n_heads, n_kv, head_dim)
    return qkv[:, : n_heads * head_dim].view(qkv.shape[0], n_heads, head_dim)


def paged_attention(
    q: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    block_tables: torch.Tensor,
    tok_seq: torch.Tensor,
    ctx_lens: torch.Tensor,
    scale: float,
    max_blocks: Optional[int] = None,
) -> torch.Tensor:
    t, nh, hd = q.shape
    if hd != 128 or k_cache.shape[2] != 16:
        # shapes outside the kernel's specialisation (test-sized models): reference path
        from . import torch_ops

        return torch_ops.paged_attention(q, k_cache, v_cache, block_tables, tok_seq, ctx_lens, scale, max_blocks)
    out = torch.empty((t, nh * hd), dtype=q.dtype, device=q.device)
    native().paged_attention(q, k_cache, v_cache, block_tables, tok_seq, ctx_lens, out, scale)
    return out


def greedy_sample(logits: torch.Tensor) -> torch.Tensor:
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    native().argmax(logits, out)
    return out
