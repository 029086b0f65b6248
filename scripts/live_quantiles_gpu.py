#!/usr/bin/env python3
"""A load run against the native engine with --live-quantiles on the GPU: the SSE parser feeds
the ring buffer, every 100 ms the delta goes through stream_hist_kernel.  Under
`rocprofv3 --kernel-trace` the trace shows the histogram kernel between the decode kernels of
the load phase; scripts/live_trace_check.py counts that from the trace CSV."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.loadgen.live import LiveQuantiles
from kvmini.loadgen.protocols import RequestSpec
from kvmini.loadgen.runner import LoadConfig, run_load_sync
from kvmini.server.native import start_native_in_thread

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
eng = LLMEngine(EngineConfig(model=model, device="cuda:0", max_num_seqs=32, max_batched_tokens=256, max_model_len=1024, kernels="hip", kv_memory_fraction=0.2))
eng.warmup()
_ns, url, stop = start_native_in_thread(eng, model_name=model)
try:
    live = LiveQuantiles(interval_s=0.1, device="cuda", verbose=True)
    out_dir = os.path.join(ROOT, "gpurun_out", "live_run")
    res = run_load_sync(LoadConfig(url=url, run_dir=out_dir, requests=200, concurrency=16, spec=RequestSpec(model=model, max_tokens=48, stream=True), quiet=True, live=live))
    from kvmini.analysis.reduce import inter_token_latency_metrics

    off = inter_token_latency_metrics(out_dir, engine="python")
    fin = json.load(open(os.path.join(out_dir, "live_itl.json")))
    print(json.dumps({"live": {k: fin[k] for k in ("itl_p50_ms", "itl_p95_ms", "itl_p99_ms", "itl_samples", "updates", "engine")},
                      "offline": {k: off[k] for k in ("itl_p50_ms", "itl_p95_ms", "itl_p99_ms", "itl_samples")}, "ok_requests": sum(r.status == 200 for r in res)}))
finally:
    stop()
