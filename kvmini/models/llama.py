"""Llama-family decoder (Llama-2/3, Mistral, CodeLlama shapes) for the native engine.

The reference never touches model code: it benchmarks whatever a KServe runtime
serves and names models only in configs (``reference-matrix.yaml:24-34``,
``profiles/tensorrt-llm/*.yaml``, ``README.md:66-72``).  Producing the headline
numbers on an MI355X box with no vLLM and no network needs a model of the same
architecture with random-init weights, so this module holds the architecture
presets and a weight container laid out for the engine's kernels:

* ``wqkv``   [(n_heads + 2 n_kv) * head_dim, hidden]  — fused Q/K/V projection
* ``wo``     [hidden, n_heads * head_dim]
* ``wgu``    [2 * intermediate, hidden] — gate and up rows interleaved in groups
  of 8 (rows 16j..16j+7 = gate[8j..], rows 16j+8..16j+15 = up[8j..]) so every
  16-row MFMA tile of the GEMM holds matching gate/up values and a wave can apply
  SwiGLU in its epilogue on its own
* ``wdown``  [hidden, intermediate]

All matrices are ``[out, in]`` row-major (``y = x @ W.T``) in bf16; with tensor
parallelism ``wqkv``/``wgu`` are split by output rows (heads / intermediate
columns) and ``wo``/``wdown`` by input columns, followed by an all-reduce.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field, replace
from typing import Any, Dict, List, Optional

import torch

GU_GROUP = 8  # gate/up interleave granularity (rows): one 16-row MFMA tile = 8 gate rows + their 8 up rows


@dataclass(frozen=True)
class LlamaConfig:
    name: str = "tiny"
    vocab_size: int = 512
    hidden_size: int = 128
    intermediate_size: int = 256
    num_layers: int = 2
    num_heads: int = 4
    num_kv_heads: int = 2
    head_dim: int = 32
    rope_theta: float = 500000.0
    rms_eps: float = 1e-5
    max_position: int = 8192
    bos_token_id: int = 1
    eos_token_id: int = 2
    # set by pad_head_dim(): the softmax scale and the rotary width of the un-padded heads
    attn_scale: Optional[float] = None
    rope_dim: Optional[int] = None
    # Llama-3.1 / 3.2 "llama3" rotary scaling: (factor, low_freq_factor, high_freq_factor,
    # original_max_position); None = plain rotary frequencies
    rope_scaling: Optional["tuple[float, float, float, int]"] = None
    tie_word_embeddings: bool = False
    extra_eos_token_ids: "tuple[int, ...]" = ()  # further end-of-sequence ids a checkpoint declares (Llama-3 Instruct: <|eot_id|> next to <|end_of_text|>)
    qkv_bias: bool = False  # Qwen2 form: the Q / K / V projections carry a bias (no other projection does)

    @property
    def softmax_scale(self) -> float:
        return self.attn_scale if self.attn_scale is not None else 1.0 / math.sqrt(self.head_dim)

    @property
    def q_size(self) -> int:
        return self.num_heads * self.head_dim

    @property
    def kv_size(self) -> int:
        return self.num_kv_heads * self.head_dim

    def param_count(self) -> int:
        per_layer = (self.q_size + 2 * self.kv_size) * self.hidden_size + self.hidden_size * self.q_size + 3 * self.intermediate_size * self.hidden_size + 2 * self.hidden_size
        return self.num_layers * per_layer + 2 * self.vocab_size * self.hidden_size + self.hidden_size

    def weight_bytes(self, bytes_per_param: float = 2.0) -> float:
        return self.param_count() * bytes_per_param


PRESETS: Dict[str, LlamaConfig] = {
    # test-sized
    "tiny": LlamaConfig(),
    "tiny-draft": LlamaConfig(name="tiny-draft", num_layers=1),
    "tiny-qkv-bias": LlamaConfig(name="tiny-qkv-bias", qkv_bias=True),
    # smallest shape that exercises every HIP kernel specialisation (head_dim 128)
    "small-hd128": LlamaConfig("small-hd128", 1024, 256, 512, 2, 4, 2, 128, 500000.0, 1e-5, 2048, 1, 2),
    # four kv heads: the smallest shape that shards over four tensor-parallel ranks
    "small-hd128-kv4": LlamaConfig("small-hd128-kv4", 1024, 256, 512, 2, 8, 4, 128, 500000.0, 1e-5, 2048, 1, 2),
    "small-hd128-qkv-bias": LlamaConfig("small-hd128-qkv-bias", 1024, 256, 512, 2, 4, 2, 128, 500000.0, 1e-5, 2048, 1, 2, qkv_bias=True),
    "small-hd128-draft": LlamaConfig("small-hd128-draft", 1024, 256, 512, 1, 4, 2, 128, 500000.0, 1e-5, 2048, 1, 2),
    # smallest shape the persistent decode-layers kernel serves (head groups of 4, K multiples of 256)
    "mid-hd128": LlamaConfig("mid-hd128", 1024, 1024, 2048, 3, 8, 2, 128, 500000.0, 1e-5, 4096, 1, 2),
    # Llama 3
    "llama-3-8b": LlamaConfig("llama-3-8b", 128256, 4096, 14336, 32, 32, 8, 128, 500000.0, 1e-5, 8192, 128000, 128001),
    "llama-3-70b": LlamaConfig("llama-3-70b", 128256, 8192, 28672, 80, 64, 8, 128, 500000.0, 1e-5, 8192, 128000, 128001),
    # Llama 3.1 / 3.2: same layers, 128 k positions through the "llama3" rotary scaling
    "llama-3.1-8b": LlamaConfig("llama-3.1-8b", 128256, 4096, 14336, 32, 32, 8, 128, 500000.0, 1e-5, 131072, 128000, 128001, rope_scaling=(8.0, 1.0, 4.0, 8192)),
    "llama-3.1-70b": LlamaConfig("llama-3.1-70b", 128256, 8192, 28672, 80, 64, 8, 128, 500000.0, 1e-5, 131072, 128000, 128001, rope_scaling=(8.0, 1.0, 4.0, 8192)),
    "llama-3.2-1b": LlamaConfig("llama-3.2-1b", 128256, 2048, 8192, 16, 32, 8, 64, 500000.0, 1e-5, 131072, 128000, 128001, rope_scaling=(32.0, 1.0, 4.0, 8192), tie_word_embeddings=True),
    "llama-3.2-3b": LlamaConfig("llama-3.2-3b", 128256, 3072, 8192, 28, 24, 8, 128, 500000.0, 1e-5, 131072, 128000, 128001, rope_scaling=(32.0, 1.0, 4.0, 8192), tie_word_embeddings=True),
    # Llama 2 / CodeLlama / Mistral (names used by the reference's matrices and engine profiles)
    "llama-2-7b": LlamaConfig("llama-2-7b", 32000, 4096, 11008, 32, 32, 32, 128, 10000.0, 1e-5, 4096, 1, 2),
    "llama-2-13b": LlamaConfig("llama-2-13b", 32000, 5120, 13824, 40, 40, 40, 128, 10000.0, 1e-5, 4096, 1, 2),
    "codellama-7b": LlamaConfig("codellama-7b", 32016, 4096, 11008, 32, 32, 32, 128, 1000000.0, 1e-5, 16384, 1, 2),
    "mistral-7b": LlamaConfig("mistral-7b", 32000, 4096, 14336, 32, 32, 8, 128, 10000.0, 1e-5, 32768, 1, 2),
    # Qwen2 / Qwen2.5: the Llama form plus a bias on the Q / K / V projections
    "qwen2.5-7b": LlamaConfig("qwen2.5-7b", 152064, 3584, 18944, 28, 28, 4, 128, 1000000.0, 1e-6, 32768, 151643, 151645, qkv_bias=True),
    "qwen2.5-0.5b": LlamaConfig("qwen2.5-0.5b", 151936, 896, 4864, 24, 14, 2, 64, 1000000.0, 1e-6, 32768, 151643, 151645, tie_word_embeddings=True, qkv_bias=True),
    # a 2.7B dense decoder in Llama form (stand-in shape for the phi-2 engine profile)
    "phi-2.7b": LlamaConfig("phi-2.7b", 51200, 2560, 10240, 32, 32, 32, 80, 10000.0, 1e-5, 2048, 1, 2),
}

_ALIASES = {
    "meta-llama/llama-3.1-8b-instruct": "llama-3.1-8b",
    "meta-llama/llama-3.1-8b": "llama-3.1-8b",
    "meta-llama/llama-3.1-70b-instruct": "llama-3.1-70b",
    "meta-llama/meta-llama-3-8b-instruct": "llama-3-8b",
    "llama3-8b": "llama-3-8b",
    "llama3-70b": "llama-3-70b",
    "llama2-7b": "llama-2-7b",
    "llama-7b": "llama-2-7b",
    "llama2-13b": "llama-2-13b",
    "phi-2": "phi-2.7b",
}


def get_config(name: str) -> LlamaConfig:
    key = name.lower()
    key = _ALIASES.get(key, key)
    if key not in PRESETS:
        raise KeyError(f"unknown model preset {name!r}; known: {sorted(PRESETS)}")
    return PRESETS[key]


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor, group: int = GU_GROUP) -> torch.Tensor:
    """[I, H], [I, H] -> [2I, H] with rows interleaved in groups of ``group``."""
    i, h = gate.shape
    assert i % group == 0, "intermediate size must be a multiple of the interleave group"
    g = gate.view(i // group, group, h)
    u = up.view(i // group, group, h)
    return torch.stack((g, u), dim=1).reshape(2 * i, h)


def deinterleave_gate_up(gu: torch.Tensor, group: int = GU_GROUP) -> "tuple[torch.Tensor, torch.Tensor]":
    """Inverse of :func:`interleave_gate_up` on the last dimension of an activation
    ``[..., 2I]`` (or on the rows of a weight, via ``.T``)."""
    *lead, n = gu.shape
    x = gu.reshape(*lead, n // (2 * group), 2, group)
    return x[..., 0, :].reshape(*lead, n // 2), x[..., 1, :].reshape(*lead, n // 2)


@dataclass
class LayerWeights:
    attn_norm: torch.Tensor
    wqkv: torch.Tensor
    wo: torch.Tensor
    mlp_norm: torch.Tensor
    wgu: torch.Tensor
    wdown: torch.Tensor
    bqkv: Optional[torch.Tensor] = None  # [(nh + 2 nkv) * hd] bias of the fused Q / K / V projection (Qwen2 form), else None
    # optional fp8 (e4m3) copies + per-output-row scales for the quantised GEMM path
    fp8: Dict[str, Any] = field(default_factory=dict)


@dataclass
class LlamaWeights:
    cfg: LlamaConfig
    embed: torch.Tensor
    layers: List[LayerWeights]
    final_norm: torch.Tensor
    lm_head: torch.Tensor
    tp_rank: int = 0
    tp_size: int = 1

    @property
    def device(self) -> torch.device:
        return self.embed.device

    def nbytes(self) -> int:
        n = self.embed.numel() * self.embed.element_size() + self.lm_head.numel() * self.lm_head.element_size()
        for lw in self.layers:
            for t in (lw.wqkv, lw.wo, lw.wgu, lw.wdown):
                n += t.numel() * t.element_size()
        return n


def random_weights(
    cfg: LlamaConfig,
    device: torch.device | str = "cpu",
    dtype: torch.dtype = torch.bfloat16,
    seed: int = 0,
    tp_rank: int = 0,
    tp_size: int = 1,
    tail_damping: "Optional[tuple[int, float]]" = None,
) -> LlamaWeights:
    """Random-init weights (N(0, 0.02) like HF initialisation, norms = 1).  Every
    TP rank draws the *full* matrix from the same seed and keeps its shard, so a
    TP=k model equals the TP=1 model of the same seed (what the multi-process
    tests check).  Generation happens on ``device`` layer by layer, so the 16 GB of
    an 8B model never exist on the host.

    ``tail_damping=(first_layer, eps)`` multiplies the residual-branch outputs (``wo``,
    ``wdown``) of layers ``>= first_layer`` by ``eps``: a synthetic stand-in for a trained
    model whose early layers already decide most tokens, so that an early-exit draft
    (speculative decoding with ``draft_model="self:<first_layer>"``) agrees with the full
    model at a controllable rate.  The full model still streams every layer's weights."""
    assert cfg.num_heads % tp_size == 0 and cfg.num_kv_heads % tp_size == 0 and cfg.intermediate_size % (tp_size * GU_GROUP) == 0
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)

    def rnd(rows: int, cols: int) -> torch.Tensor:
        return (torch.randn(rows, cols, generator=gen, device=dev, dtype=torch.float32) * 0.02).to(dtype)

    h, hd = cfg.hidden_size, cfg.head_dim
    nh_l, nkv_l, i_l = cfg.num_heads // tp_size, cfg.num_kv_heads // tp_size, cfg.intermediate_size // tp_size

    def shard_rows(w: torch.Tensor, n_local: int) -> torch.Tensor:
        return w[tp_rank * n_local : (tp_rank + 1) * n_local].contiguous()

    def shard_cols(w: torch.Tensor, n_local: int) -> torch.Tensor:
        return w[:, tp_rank * n_local : (tp_rank + 1) * n_local].contiguous()

    embed = rnd(cfg.vocab_size, h)
    layers: List[LayerWeights] = []
    for li in range(cfg.num_layers):
        damp = tail_damping[1] if tail_damping is not None and li >= tail_damping[0] else None
        wq = shard_rows(rnd(cfg.q_size, h), nh_l * hd)
        wk = shard_rows(rnd(cfg.kv_size, h), nkv_l * hd)
        wv = shard_rows(rnd(cfg.kv_size, h), nkv_l * hd)
        wo = shard_cols(rnd(h, cfg.q_size), nh_l * hd)
        wg = shard_rows(rnd(cfg.intermediate_size, h), i_l)
        wu = shard_rows(rnd(cfg.intermediate_size, h), i_l)
        wd = shard_cols(rnd(h, cfg.intermediate_size), i_l)
        if damp is not None:
            wo, wd = (wo.float() * damp).to(dtype), (wd.float() * damp).to(dtype)
        layers.append(
            LayerWeights(
                attn_norm=torch.ones(h, device=dev, dtype=dtype),
                wqkv=torch.cat((wq, wk, wv), dim=0).contiguous(),
                wo=wo,
                mlp_norm=torch.ones(h, device=dev, dtype=dtype),
                wgu=interleave_gate_up(wg, wu).contiguous(),
                wdown=wd,
                bqkv=(torch.cat((shard_rows(rnd(cfg.q_size, 1), nh_l * hd), shard_rows(rnd(cfg.kv_size, 1), nkv_l * hd), shard_rows(rnd(cfg.kv_size, 1), nkv_l * hd))).view(-1).contiguous()
                      if cfg.qkv_bias else None),
            )
        )
        del wq, wk, wv, wg, wu
    final_norm = torch.ones(h, device=dev, dtype=dtype)
    lm_head = rnd(cfg.vocab_size, h)
    return LlamaWeights(cfg, embed, layers, final_norm, lm_head, tp_rank, tp_size)


def config_from_hf(config: "str | Dict[str, Any]", name: Optional[str] = None) -> LlamaConfig:
    """The architecture of a Hugging Face checkpoint from its ``config.json`` (a path to the file or
    its directory, or the parsed document): any decoder of the Llama form — ``model_type`` llama,
    mistral or qwen2 (Q / K / V bias); grouped-query attention, SwiGLU, RMSNorm, rotary positions (plain or "llama3" scaled),
    tied or separate output embeddings, an explicit ``head_dim``.  What the engine has no kernels for
    is refused here rather than loaded wrongly: projection biases other than Qwen2's Q / K / V bias,
    other rotary scalings (linear / dynamic / yarn), mixture-of-experts layers.  A sliding attention window (Mistral-7B
    v0.1: 4096) is honoured by limiting the context to the window, inside which it changes nothing."""
    import json
    import os

    gen: Dict[str, Any] = {}
    if isinstance(config, str):
        path = os.path.join(config, "config.json") if os.path.isdir(config) else config
        with open(path) as f:
            hf = json.load(f)
        if name is None:
            name = os.path.basename(os.path.dirname(os.path.abspath(path))) or "hf-model"
        gpath = os.path.join(os.path.dirname(os.path.abspath(path)), "generation_config.json")
        if os.path.exists(gpath):  # (chat checkpoints list their end-of-turn token here)
            with open(gpath) as f:
                gen = json.load(f)
    else:
        hf = dict(config)
    mt = hf.get("model_type", "llama")
    if mt not in ("llama", "mistral", "qwen2"):
        raise ValueError(f"model_type {mt!r} is not a Llama-form decoder this engine serves (llama, mistral, qwen2)")
    if hf.get("attention_bias") or hf.get("mlp_bias"):
        raise ValueError("projection biases beyond Qwen2's Q / K / V bias (attention_bias / mlp_bias) are not supported")
    if mt == "qwen2" and hf.get("use_sliding_window"):
        raise ValueError("Qwen2 checkpoints with use_sliding_window are not supported")
    if hf.get("hidden_act", "silu") not in ("silu", "swish"):
        raise ValueError(f"hidden_act {hf.get('hidden_act')!r} is not supported (SwiGLU only)")
    if hf.get("num_local_experts") or hf.get("num_experts"):
        raise ValueError("mixture-of-experts checkpoints are not supported")
    hidden, nh = int(hf["hidden_size"]), int(hf["num_attention_heads"])
    hd = int(hf.get("head_dim") or hidden // nh)
    rs = _hf_rope(hf)
    scaling = None
    if rs and (rs.get("rope_type") or rs.get("type")) not in (None, "default"):
        kind = rs.get("rope_type") or rs.get("type")
        if kind != "llama3":
            raise ValueError(f"rope_scaling type {kind!r} is not supported (llama3 only)")
        scaling = (float(rs.get("factor", 1.0)), float(rs.get("low_freq_factor", 1.0)), float(rs.get("high_freq_factor", 4.0)), int(rs.get("original_max_position_embeddings", 8192)))
    max_pos = int(hf.get("max_position_embeddings", 8192))
    window = hf.get("sliding_window") if mt != "qwen2" else None  # (Qwen2 carries the field but leaves it off: use_sliding_window false)
    if window and int(window) < max_pos:
        max_pos = int(window)
    rope_theta = hf.get("rope_theta", (rs or {}).get("rope_theta"))

    def first(v: Any, default: int) -> int:
        if isinstance(v, (list, tuple)):
            v = v[0] if v else None
        return default if v is None else int(v)

    return LlamaConfig(
        name=name or str(hf.get("_name_or_path") or "hf-model"),
        vocab_size=int(hf["vocab_size"]),
        hidden_size=hidden,
        intermediate_size=int(hf["intermediate_size"]),
        num_layers=int(hf["num_hidden_layers"]),
        num_heads=nh,
        num_kv_heads=int(hf.get("num_key_value_heads") or nh),
        head_dim=hd,
        rope_theta=float(rope_theta if rope_theta is not None else 10000.0),
        rms_eps=float(hf.get("rms_norm_eps", 1e-5)),
        max_position=max_pos,
        bos_token_id=first(hf.get("bos_token_id"), 1),
        eos_token_id=first(hf.get("eos_token_id"), 2),
        rope_scaling=scaling,
        tie_word_embeddings=bool(hf.get("tie_word_embeddings", False)),
        qkv_bias=mt == "qwen2",
        extra_eos_token_ids=tuple(sorted({int(t) for src in (hf.get("eos_token_id"), gen.get("eos_token_id")) for t in (src if isinstance(src, (list, tuple)) else [src]) if t is not None}
                                         - {first(hf.get("eos_token_id"), 2)})),
    )


def _hf_rope(hf: Dict[str, Any]) -> Optional[Dict[str, Any]]:
    """The rotary settings of a ``config.json``: ``rope_scaling`` (with ``rope_theta`` beside it) in
    the older layout, one ``rope_parameters`` object (theta, type and factors together) in the newer."""
    rs = hf.get("rope_scaling")
    if not rs and isinstance(hf.get("rope_parameters"), dict):
        rs = hf["rope_parameters"]
    return rs if isinstance(rs, dict) else None


def resolve_model(model: str, weights_path: Optional[str] = None) -> "tuple[LlamaConfig, Optional[str]]":
    """``model`` is a preset name (``llama-3-8b``) or a Hugging Face checkpoint directory: the
    architecture then comes from its ``config.json`` and, unless ``weights_path`` says otherwise,
    the weights from its ``*.safetensors``.  With a preset name and a ``weights_path`` the
    checkpoint's ``config.json`` is checked against the preset when the weights are loaded."""
    import glob
    import os

    if os.path.isdir(model) and os.path.exists(os.path.join(model, "config.json")):
        cfg = config_from_hf(model)
        if weights_path is None and glob.glob(os.path.join(model, "*.safetensors")):
            weights_path = model
        return cfg, weights_path
    return get_config(model), weights_path


def load_hf_safetensors(path: str, cfg: LlamaConfig, device: torch.device | str = "cpu", dtype: torch.dtype = torch.bfloat16, tp_rank: int = 0, tp_size: int = 1) -> LlamaWeights:
    """Load a Hugging Face Llama checkpoint directory (``*.safetensors``) into the fused layout.
    Only safetensors is read — nothing is unpickled.  The files are memory-mapped and read one
    layer at a time, each tensor-parallel rank only its own rows / columns, so the host never
    holds more than a layer of a 70B checkpoint."""
    import glob
    import os

    from safetensors import safe_open

    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no .safetensors files under {path}")
    if cfg.num_heads % tp_size or cfg.num_kv_heads % tp_size or cfg.intermediate_size % (tp_size * GU_GROUP):
        raise ValueError(f"{cfg.name}: {cfg.num_heads} heads / {cfg.num_kv_heads} kv heads / intermediate {cfg.intermediate_size} do not split over {tp_size} tensor-parallel ranks")
    check_hf_config(os.path.join(path, "config.json"), cfg)
    handles = [safe_open(fpath, framework="pt", device="cpu") for fpath in files]
    where: Dict[str, Any] = {}
    for h in handles:
        for k in h.keys():
            where[k] = h
    dev = torch.device(device)
    hd = cfg.rope_dim or cfg.head_dim  # the checkpoint's head width (heads are zero-padded afterwards when the kernels want them wider)
    nh_l, nkv_l, i_l = cfg.num_heads // tp_size, cfg.num_kv_heads // tp_size, cfg.intermediate_size // tp_size

    def need(name: str) -> Any:
        if name not in where:
            raise KeyError(f"{path}: tensor {name!r} is missing from the checkpoint")
        return where[name]

    def rows(name: str, n: int) -> torch.Tensor:  # this rank's n output rows
        return need(name).get_slice(name)[tp_rank * n : (tp_rank + 1) * n].to(dtype)

    def cols(name: str, n: int) -> torch.Tensor:  # this rank's n input columns
        return need(name).get_slice(name)[:, tp_rank * n : (tp_rank + 1) * n].to(dtype)

    def whole(name: str) -> torch.Tensor:
        return need(name).get_tensor(name).to(dtype)

    has_bias = any(k.endswith("self_attn.q_proj.bias") for k in where)
    if has_bias != cfg.qkv_bias:
        raise ValueError(f"{path}: the checkpoint {'has' if has_bias else 'lacks'} Q / K / V biases but the architecture {cfg.name!r} says otherwise")
    if any(k.endswith("self_attn.o_proj.bias") or k.endswith("mlp.down_proj.bias") for k in where):
        raise ValueError(f"{path}: biases on the output / MLP projections are not supported")

    def vec(name: str, n: int) -> torch.Tensor:  # this rank's n entries of a bias
        return need(name).get_slice(name)[tp_rank * n : (tp_rank + 1) * n].to(dtype)
    layers = []
    for li in range(cfg.num_layers):
        p = f"model.layers.{li}."
        wqkv = torch.cat((rows(p + "self_attn.q_proj.weight", nh_l * hd), rows(p + "self_attn.k_proj.weight", nkv_l * hd), rows(p + "self_attn.v_proj.weight", nkv_l * hd)), 0)
        wgu = interleave_gate_up(rows(p + "mlp.gate_proj.weight", i_l).contiguous(), rows(p + "mlp.up_proj.weight", i_l).contiguous())
        bqkv = None
        if has_bias:
            bqkv = torch.cat((vec(p + "self_attn.q_proj.bias", nh_l * hd), vec(p + "self_attn.k_proj.bias", nkv_l * hd), vec(p + "self_attn.v_proj.bias", nkv_l * hd))).contiguous().to(dev)
        layers.append(
            LayerWeights(
                bqkv=bqkv,
                attn_norm=whole(p + "input_layernorm.weight").to(dev),
                wqkv=wqkv.contiguous().to(dev),
                wo=cols(p + "self_attn.o_proj.weight", nh_l * hd).contiguous().to(dev),
                mlp_norm=whole(p + "post_attention_layernorm.weight").to(dev),
                wgu=wgu.contiguous().to(dev),
                wdown=cols(p + "mlp.down_proj.weight", i_l).contiguous().to(dev),
            )
        )
    embed = whole("model.embed_tokens.weight").to(dev)
    lm_head = embed if "lm_head.weight" not in where else whole("lm_head.weight").to(dev)
    return LlamaWeights(cfg, embed, layers, whole("model.norm.weight").to(dev), lm_head, tp_rank, tp_size)


def check_hf_config(config_json: str, cfg: LlamaConfig) -> None:
    """Compare a checkpoint's ``config.json`` (when present) with the architecture preset the
    engine was started with; a mismatch would load without an error and produce wrong logits."""
    import json
    import os

    if not os.path.exists(config_json):
        return
    with open(config_json) as f:
        hf = json.load(f)
    nh = hf.get("num_attention_heads")
    want = {
        "hidden_size": cfg.hidden_size, "intermediate_size": cfg.intermediate_size, "num_hidden_layers": cfg.num_layers, "num_attention_heads": cfg.num_heads,
        "num_key_value_heads": cfg.num_kv_heads, "vocab_size": cfg.vocab_size, "rms_norm_eps": cfg.rms_eps, "rope_theta": cfg.rope_theta,
    }
    bad = [f"{k}: checkpoint {hf[k]!r}, preset {v!r}" for k, v in want.items() if k in hf and (abs(float(hf[k]) - float(v)) > 1e-9 * max(1.0, abs(float(v))))]
    if "head_dim" in hf or (nh and "hidden_size" in hf):
        hd = hf.get("head_dim") or hf["hidden_size"] // nh
        if hd != (cfg.rope_dim or cfg.head_dim):
            bad.append(f"head_dim: checkpoint {hd}, preset {cfg.rope_dim or cfg.head_dim}")
    if (hf.get("model_type") == "qwen2") != cfg.qkv_bias:
        bad.append(f"Q / K / V bias: checkpoint model_type {hf.get('model_type')!r}, preset qkv_bias={cfg.qkv_bias}")
    if "tie_word_embeddings" in hf and bool(hf["tie_word_embeddings"]) != cfg.tie_word_embeddings:
        bad.append(f"tie_word_embeddings: checkpoint {hf['tie_word_embeddings']}, preset {cfg.tie_word_embeddings}")
    rs = _hf_rope(hf)
    if "rope_theta" not in hf and rs and "rope_theta" in rs and abs(float(rs["rope_theta"]) - cfg.rope_theta) > 1e-9 * max(1.0, cfg.rope_theta):
        bad.append(f"rope_theta: checkpoint {rs['rope_theta']!r}, preset {cfg.rope_theta!r}")
    if rs and (rs.get("rope_type") or rs.get("type")) not in (None, "default"):
        kind = rs.get("rope_type") or rs.get("type")
        got = (float(rs.get("factor", 1.0)), float(rs.get("low_freq_factor", 1.0)), float(rs.get("high_freq_factor", 4.0)), int(rs.get("original_max_position_embeddings", 8192)))
        if kind != "llama3":
            bad.append(f"rope_scaling type {kind!r} is not supported (llama3 only)")
        elif cfg.rope_scaling is None or any(abs(a - b) > 1e-9 for a, b in zip(got, cfg.rope_scaling)):
            bad.append(f"rope_scaling: checkpoint {got}, preset {cfg.rope_scaling}")
    elif cfg.rope_scaling is not None and ("rope_scaling" in hf or "rope_parameters" in hf):
        bad.append(f"rope_scaling: checkpoint has none, preset {cfg.rope_scaling}")
    if bad:
        raise ValueError(f"checkpoint config.json does not match preset {cfg.name!r}: " + "; ".join(bad))


def rope_tables(cfg: LlamaConfig, max_pos: int, device: torch.device | str) -> "tuple[torch.Tensor, torch.Tensor]":
    """cos/sin tables [max_pos, head_dim/2] in fp32 (rotate-half / NeoX pairing:
    element i pairs with i + head_dim/2, as in HF Llama)."""
    half = cfg.head_dim // 2
    real = (cfg.rope_dim or cfg.head_dim) // 2  # padded heads: pairs beyond the real width are (0, 0) and do not turn
    inv = torch.zeros(half, dtype=torch.float64)
    inv[:real] = 1.0 / (cfg.rope_theta ** (torch.arange(0, real, dtype=torch.float64) / real))
    if cfg.rope_scaling is not None:
        # "llama3" scaling: wavelengths beyond the original context are slowed by `factor`,
        # short ones are kept, the band in between is blended — it changes the low-frequency
        # components at EVERY position, so checkpoints trained with it need it from position 0
        factor, low_f, high_f, orig = cfg.rope_scaling
        wavelen = 2.0 * math.pi / inv[:real]
        smooth = ((orig / wavelen) - low_f) / (high_f - low_f)
        blended = (1.0 - smooth) * inv[:real] / factor + smooth * inv[:real]
        scaled = torch.where(wavelen > orig / low_f, inv[:real] / factor, torch.where(wavelen < orig / high_f, inv[:real], blended))
        inv[:real] = scaled
    ang = torch.arange(max_pos, dtype=torch.float64)[:, None] * inv[None, :]
    return ang.cos().float().to(device), ang.sin().float().to(device)


def pad_head_dim(weights: "LlamaWeights", to: int = 128) -> "LlamaWeights":
    """The same model with every attention head widened to ``to`` dims by zero rows of the
    Q/K/V projections and zero columns of the output projection — head widths such as 80 or 96
    then run on the kernels specialised for 128.  The rotate-half halves of a head move to
    ``[0, hd/2)`` and ``[to/2, to/2 + hd/2)`` so that the pairing ``i <-> i + to/2`` is the
    original ``i <-> i + hd/2``; scores, softmax scale (``attn_scale``) and outputs are
    unchanged, the KV cache grows by ``to / hd``."""
    cfg = weights.cfg
    hd = cfg.head_dim
    if hd == to:
        return weights
    assert hd < to and hd % 2 == 0 and to % 2 == 0
    h2, t2 = hd // 2, to // 2

    def widen_rows(w: torch.Tensor) -> torch.Tensor:  # [heads * hd, H] -> [heads * to, H]
        heads = w.shape[0] // hd
        src = w.view(heads, hd, w.shape[1])
        out = torch.zeros((heads, to, w.shape[1]), dtype=w.dtype, device=w.device)
        out[:, :h2] = src[:, :h2]
        out[:, t2 : t2 + h2] = src[:, h2:]
        return out.view(heads * to, w.shape[1])

    layers = []
    for lw in weights.layers:
        assert not lw.fp8, "pad before quantising"
        bq = None if lw.bqkv is None else widen_rows(lw.bqkv[:, None]).view(-1).contiguous()
        layers.append(replace(lw, wqkv=widen_rows(lw.wqkv).contiguous(), wo=widen_rows(lw.wo.T.contiguous()).T.contiguous(), bqkv=bq))
    new_cfg = replace(cfg, head_dim=to, attn_scale=cfg.softmax_scale, rope_dim=cfg.rope_dim or hd)
    return replace(weights, cfg=new_cfg, layers=layers)


def quantize_fp8_rows(w: torch.Tensor) -> "tuple[torch.Tensor, torch.Tensor]":
    """Per-output-row symmetric quantisation to OCP e4m3 (gfx950's native FP8 —
    not the FNUZ encoding of the previous generation).  Returns (w_fp8, scale[rows])."""
    amax = w.float().abs().amax(dim=1).clamp_min(1e-12)
    scale = amax / 448.0
    q = (w.float() / scale[:, None]).to(torch.float8_e4m3fn)
    return q, scale.float()


E2M1_VALUES = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)  # magnitudes of the 4-bit float, code = index (+ 8 for negative)
MX_BLOCK = 32


def quantize_mxfp4(w: torch.Tensor) -> "tuple[torch.Tensor, torch.Tensor]":
    """OCP microscaling FP4 (MXFP4) of a ``[N, K]`` weight, blocks of 32 along K: one E8M0
    scale 2^(floor(log2(amax)) - 2) per block (2 = the exponent of e2m1's largest value, 6 =
    1.5 * 2^2), elements rounded to the nearest e2m1 value and saturated at +-6.  Returns
    (codes uint8 ``[N, K]`` in 0..15, scale bytes uint8 ``[N, K/32]``)."""
    n, k = w.shape
    if k % MX_BLOCK:
        raise ValueError(f"in_features {k} must be a multiple of {MX_BLOCK} for MXFP4")
    x = w.float().view(n, k // MX_BLOCK, MX_BLOCK)
    amax = x.abs().amax(-1)
    e = torch.floor(torch.log2(amax.clamp_min(2.0 ** -120))) - 2.0
    e = e.clamp(-127.0, 127.0)
    y = x / torch.exp2(e)[..., None]
    mids = torch.tensor([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0], device=w.device)
    mag = torch.bucketize(y.abs(), mids)  # index of the nearest magnitude; beyond 5 -> 6 (saturation)
    codes = (mag + 8 * (y < 0)).to(torch.uint8).view(n, k)
    return codes, (e + 127.0).to(torch.uint8)


def dequantize_mxfp4(codes: torch.Tensor, scales: torch.Tensor, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    n, k = codes.shape
    lut = torch.tensor(E2M1_VALUES + tuple(-v for v in E2M1_VALUES), device=codes.device)
    vals = lut[codes.long()].view(n, k // MX_BLOCK, MX_BLOCK) * torch.exp2(scales.float() - 127.0)[..., None]
    return vals.view(n, k).to(dtype)


def fake_quantize_mxfp4(weights: "LlamaWeights") -> None:
    """Round every layer projection through MXFP4 and back (in place): the numerics of 4-bit
    weight-only quantisation for the reference ops, and the oracle of the MXFP4 kernel tests."""
    if getattr(weights, "quantization", None) == "mxfp4":
        return
    for lw in weights.layers:
        for name in ("wqkv", "wo", "wgu", "wdown"):
            w = getattr(lw, name)
            if w.shape[1] % MX_BLOCK == 0:
                setattr(lw, name, dequantize_mxfp4(*quantize_mxfp4(w), dtype=w.dtype))
    weights.quantization = "mxfp4"


def fake_quantize_fp8(weights: "LlamaWeights") -> None:
    """Round every layer projection through per-row e4m3 and back (in place): the
    numerics of weight-only FP8 for the reference (non-HIP) ops, and the oracle the
    FP8 kernel tests compare against.  The LM head is left in full precision."""
    if getattr(weights, "quantization", None) == "fp8":
        return
    for lw in weights.layers:
        for name in ("wqkv", "wo", "wgu", "wdown"):
            w = getattr(lw, name)
            q, scale = quantize_fp8_rows(w)
            setattr(lw, name, (q.float() * scale[:, None]).to(w.dtype))
    weights.quantization = "fp8"


def replicate_kv_heads(weights: "LlamaWeights", factor: int) -> "LlamaWeights":
    """The same model with every K/V head stored ``factor`` times (rows of the K and V
    projections duplicated), i.e. ``factor`` times fewer query heads per kv head — a grouping
    ratio the attention kernels are not specialised for (5, 16, ...) becomes one they are, at
    the price of a ``factor`` times larger KV cache."""
    cfg = weights.cfg
    if factor == 1:
        return weights
    nh_l, nkv_l, hd = cfg.num_heads // weights.tp_size, cfg.num_kv_heads // weights.tp_size, cfg.head_dim
    assert (nh_l // nkv_l) % factor == 0
    layers = []
    for lw in weights.layers:
        assert not lw.fp8, "replicate before quantising"
        q, k, v = lw.wqkv.split((nh_l * hd, nkv_l * hd, nkv_l * hd), 0)
        k = k.view(nkv_l, 1, hd, -1).expand(nkv_l, factor, hd, k.shape[1]).reshape(nkv_l * factor * hd, -1)
        v = v.view(nkv_l, 1, hd, -1).expand(nkv_l, factor, hd, v.shape[1]).reshape(nkv_l * factor * hd, -1)
        bq = lw.bqkv
        if bq is not None:
            bq_q, bq_k, bq_v = bq.split((nh_l * hd, nkv_l * hd, nkv_l * hd), 0)
            bq_k = bq_k.view(nkv_l, 1, hd).expand(nkv_l, factor, hd).reshape(-1)
            bq_v = bq_v.view(nkv_l, 1, hd).expand(nkv_l, factor, hd).reshape(-1)
            bq = torch.cat((bq_q, bq_k, bq_v)).contiguous()
        layers.append(replace(lw, wqkv=torch.cat((q, k, v), 0).contiguous(), bqkv=bq))
    return replace(weights, cfg=replace(cfg, num_kv_heads=cfg.num_kv_heads * factor), layers=layers)
