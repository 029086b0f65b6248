#!/usr/bin/env python3
"""Where a layer's time goes inside the persistent decode-layers kernel: workgroup 0's stamps of
the 100 MHz real-time counter at every phase boundary, median over the layers of one decode step
(Llama-3-8B, batch 20 by default).  Run with KVMINI_DECODE_LAYERS_TIMELINE=1."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KVMINI_DECODE_LAYERS_TIMELINE", "1")
os.environ.setdefault("KVMINI_DECODE_LAYERS", "1")
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = LLMEngine(EngineConfig(model=model, device="cuda:0", max_num_seqs=32, max_batched_tokens=256, max_model_len=2048, kernels="hip", kv_memory_fraction=0.2))
eng.warmup()
prompts = [[1] + list(range(100 + i, 113 + i)) for i in range(batch)]
seqs = [eng.make_sequence(p, SamplingParams(max_tokens=24, ignore_eos=True)) for p in prompts]
for s in seqs:
    eng.scheduler.add(s)
for _ in range(12):
    eng.step()
torch.cuda.synchronize()
dl = eng.runner.decode_layers
L = eng.runner.cfg.num_layers
tl = dl.read_timeline(L).double() / 100.0  # microseconds
names = {(1, 2): "residual update (after down)", (2, 3): "QKV: own share", (3, 4): "barrier after QKV", (4, 5): "attention: own items", (5, 6): "barrier after attention",
         (6, 7): "output projection: own share", (7, 8): "barrier after output projection", (8, 9): "residual update (after o)", (9, 10): "gate/up: own share",
         (10, 11): "barrier after gate/up", (11, 12): "down: own share"}
rows = {}
for (a, b), name in names.items():
    vals = [float(tl[li, b] - tl[li, a]) for li in range(1, L)]
    rows[name] = round(statistics.median(vals), 2)
per_layer = [float(tl[li + 1, 0] - tl[li, 0]) for li in range(1, L - 1)]
rows["barrier after down (next layer's stamp 1 - stamp 12)"] = round(statistics.median([float(tl[li + 1, 1] - tl[li, 12]) for li in range(1, L - 1)]), 2)
out = {"bench": "decode_layers_timeline", "model": model, "batch": batch, "us_per_layer_median": round(statistics.median(per_layer), 2), "phases_us": rows, "error_word": dl.error_word()}
print(json.dumps(out, indent=1))
