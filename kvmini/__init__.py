"""kvmini — MI355X-native LLM-serving benchmark harness and reference engine.

One importable package (the reference is a set of flat scripts glued with
``subprocess``; see reference ``kvmini/cli.py:41-117``).  Sub-packages:

* ``loadgen``   asyncio load generator, SSE/TTFT parser, OTLP-JSON tracing
* ``analysis``  requests.csv -> results.json reduction (quantiles, histograms, cold/warm)
* ``telemetry`` amd-smi sampler, energy integration, ROCm-exporter Prometheus queries
* ``cost``      resource-seconds cost estimator, legacy calculator, capacity planner
* ``cluster``   pluggable cluster backends (kubectl / local process / fake)
* ``deploy``    structured InferenceService manifest builders (vLLM-ROCm, TGI-ROCm)
* ``report``    HTML reports
* ``gates``     SLO gate, canary comparator, config validator
* ``harness``   sweeps, A/B compare, chaos, fairness, parity, cache probe, quality, bundle
* ``server``    built-in OpenAI-compatible servers (CPU mock and native MI355X engine)
* ``engine``    continuous-batching Llama engine (paged KV, HIP graphs, speculative decode)
* ``models``    model architecture configs and weight containers
* ``ops``       Python bindings of the hand-written gfx950 HIP kernels
* ``parallel``  RCCL (torch.distributed) tensor/data parallel and telemetry all-gather
"""

__version__ = "0.1.0"
