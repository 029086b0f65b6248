"""Engine ops backed by the gfx950 HIP kernels (same signatures as ``torch_ops``).

GEMMs with at most 64 token rows — every decode step and most mixed
decode+prefill steps — go to the weight-streaming MFMA kernel
(``csrc/skinny_gemm.hip``); larger prefill chunks use the library GEMM
(hipBLASLt through ``torch``), which is the right tool once the op is
compute-bound.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import gpu_available, native

SKINNY_MAX_M = 64
# set by the runner when it re-orders weights into MFMA fragment order
PACKED = False
WAVES = 0  # 0 = kernel picks 4 or 8 wavefronts per workgroup from K


def require() -> None:
    """Fail loudly when the HIP path cannot run (missing build or no GPU)."""
    native()
    if not gpu_available():
        raise RuntimeError("kvmini HIP kernels need a ROCm GPU (torch.cuda.is_available() is False)")


def add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, w: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    y = torch.empty_like(x)
    native().add_rmsnorm(x, residual, w, y, eps)  # residual updated in place
    return y, residual


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    y, _ = add_rmsnorm(x, torch.zeros_like(x), w, eps)
    return y


def linear(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    m = x.shape[0]
    if m <= SKINNY_MAX_M and m % 16 == 0 and x.dtype == torch.bfloat16:
        y = torch.empty((m, w.shape[0]), dtype=x.dtype, device=x.device)
        native().skinny_gemm(w, x, y, False, PACKED, WAVES)
        return y
    if PACKED:
        y = torch.empty((m, w.shape[0]), dtype=x.dtype, device=x.device)
        for r0 in range(0, m, SKINNY_MAX_M):
            r1 = min(m, r0 + SKINNY_MAX_M)
            native().skinny_gemm(w, x[r0:r1], y[r0:r1], False, True, WAVES)
        return y
    return F.linear(x, w)


def linear_swiglu(x: torch.Tensor, wgu: torch.Tensor) -> torch.Tensor:
    m = x.shape[0]
    inter = wgu.shape[0] // 2
    act = torch.empty((m, inter), dtype=x.dtype, device=x.device)
    if m <= SKINNY_MAX_M and m % 16 == 0 and x.dtype == torch.bfloat16:
        native().skinny_gemm(wgu, x, act, True, PACKED, WAVES)
        return act
    if PACKED:
        for r0 in range(0, m, SKINNY_MAX_M):
            r1 = min(m, r0 + SKINNY_MAX_M)
            native().skinny_gemm(wgu, x[r0:r1], act[r0:r1], True, True, WAVES)
        return act
    gu = F.linear(x, wgu)
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
    """Rotates q in place inside ``qkv`` and returns it as a strided [T, nh, hd] view."""
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
