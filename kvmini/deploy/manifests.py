"""Structured InferenceService manifest builders for ROCm runtimes.

The reference patches a YAML file with line-anchored ``sed`` (``deploy.sh:91-99``,
repeated in ``sweeps/mig-sweep.sh:94-97``), which cannot set GPU counts or args and
leaves labels stale; its backend adapters are heredocs with NVIDIA resources
(``runners/backends/*/deploy.sh``).  These builders return plain dicts:

* :func:`vllm_rocm_isvc` — KServe ``model`` predictor on a vLLM-ROCm ServingRuntime
  (``modelFormat: vllm``; ``amd.com/gpu`` = tensor-parallel size; vLLM args from the
  profile's ``vllm_features``; ``/dev/kfd`` + ``/dev/dri`` are injected by the AMD
  device plugin, shared memory volume for RCCL);
* :func:`tgi_rocm_isvc` — custom-container predictor running
  ``ghcr.io/huggingface/text-generation-inference:*-rocm`` (``NUM_SHARD``);
* :func:`native_isvc` — this repository's engine image (``kvmini serve``);
* :func:`apply_partition_profile` — MI355X compute-partition / SR-IOV profiles
  (resource key, node selector) in either of the two schemas the reference ships
  for MIG (``profiles/mig/*.yaml`` vs ``runners/profiles/mig/*.yaml``);
* :func:`apply_autoscaling` — Knative annotations for the autoscale sweep.
"""

from __future__ import annotations

import copy
from typing import Any, Dict, List, Optional

from kvmini.utils import dns1123_label

GPU_RESOURCE = "amd.com/gpu"
VLLM_ROCM_IMAGE = "rocm/vllm:latest"
TGI_ROCM_IMAGE = "ghcr.io/huggingface/text-generation-inference:latest-rocm"
NATIVE_IMAGE = "kvmini/native-engine:latest"

_DEFAULT_ANNOTATIONS = {
    "serving.kserve.io/deploymentMode": "Serverless",
    "autoscaling.knative.dev/class": "kpa.autoscaling.knative.dev",
    "autoscaling.knative.dev/metric": "concurrency",
    "autoscaling.knative.dev/target": "10",
    "autoscaling.knative.dev/scaleToZeroGracePeriod": "30s",
    "serving.kserve.io/enable-metrics": "true",
}


def _meta(name: str, namespace: str, backend: str, annotations: Optional[Dict[str, str]] = None) -> Dict[str, Any]:
    name = dns1123_label(name)
    ann = dict(_DEFAULT_ANNOTATIONS)
    ann.update(annotations or {})
    return {"name": name, "namespace": namespace, "labels": {"app": name, "kvmini/backend": backend}, "annotations": ann}


def _resources(gpus: int, cpu: str, memory: str, gpu_key: str = GPU_RESOURCE) -> Dict[str, Any]:
    limits: Dict[str, str] = {"memory": memory}
    requests: Dict[str, str] = {"cpu": cpu, "memory": memory}
    if gpus > 0:
        limits[gpu_key] = str(gpus)
        requests[gpu_key] = str(gpus)
    return {"limits": limits, "requests": requests}


def _env(d: Dict[str, Any]) -> List[Dict[str, str]]:
    return [{"name": k, "value": str(v)} for k, v in d.items() if v is not None]


def vllm_rocm_isvc(
    name: str,
    namespace: str = "default",
    model_uri: Optional[str] = None,
    model_name: Optional[str] = None,
    runtime: str = "kserve-vllm-rocm",
    tensor_parallel_size: int = 1,
    pipeline_parallel_size: int = 1,
    max_model_len: int = 8192,
    gpu_memory_utilization: float = 0.90,
    vllm_args: Optional[List[str]] = None,
    extra_env: Optional[Dict[str, Any]] = None,
    cpu: str = "8",
    memory: str = "64Gi",
    min_replicas: int = 0,
    max_replicas: int = 4,
    annotations: Optional[Dict[str, str]] = None,
    device: str = "rocm",
) -> Dict[str, Any]:
    gpus = 0 if device == "cpu" else tensor_parallel_size * pipeline_parallel_size
    args = ["--max-model-len", str(max_model_len), "--disable-log-requests"]
    if device != "cpu":
        args += ["--gpu-memory-utilization", str(gpu_memory_utilization), "--tensor-parallel-size", str(tensor_parallel_size)]
        if pipeline_parallel_size > 1:
            args += ["--pipeline-parallel-size", str(pipeline_parallel_size)]
    seen = set(a for a in args if a.startswith("--"))
    va = list(vllm_args or [])
    i = 0
    while i < len(va):  # profile args override the defaults above
        flag = va[i]
        takes_val = i + 1 < len(va) and not va[i + 1].startswith("--")
        if flag in seen:
            j = args.index(flag)
            del args[j : j + (2 if j + 1 < len(args) and not args[j + 1].startswith("--") else 1)]
        args += va[i : i + (2 if takes_val else 1)]
        i += 2 if takes_val else 1
    env = {
        "MODEL_NAME": model_name or name,
        "TENSOR_PARALLEL_SIZE": tensor_parallel_size,
        "PIPELINE_PARALLEL_SIZE": pipeline_parallel_size,
        "GPU_MEMORY_UTILIZATION": gpu_memory_utilization,
        "MAX_MODEL_LEN": max_model_len,
        "VLLM_ARGS": " ".join(args),
        # ROCm specifics: RCCL over xGMI needs dmabuf IPC on recent kernels; aiter kernels on gfx950
        "HSA_ENABLE_IPC_MODE_LEGACY": "0",
        "VLLM_ROCM_USE_AITER": "1",
    }
    env.update(extra_env or {})
    ann = {"autoscaling.knative.dev/minScale": str(min_replicas), "autoscaling.knative.dev/maxScale": str(max_replicas)}
    ann.update(annotations or {})
    model: Dict[str, Any] = {
        "modelFormat": {"name": "vllm"},
        "runtime": runtime,
        "args": args,
        "env": _env(env),
        "resources": _resources(gpus, cpu, memory),
    }
    if model_uri:
        model["storageUri"] = model_uri
    predictor: Dict[str, Any] = {"model": model}
    if gpus > 1:
        # RCCL uses /dev/shm for intra-node bootstrap buffers
        predictor["volumes"] = [{"name": "dshm", "emptyDir": {"medium": "Memory", "sizeLimit": "16Gi"}}]
        model["volumeMounts"] = [{"name": "dshm", "mountPath": "/dev/shm"}]
    return {"apiVersion": "serving.kserve.io/v1beta1", "kind": "InferenceService", "metadata": _meta(name, namespace, "vllm-rocm", ann), "spec": {"predictor": predictor}}


def tgi_rocm_isvc(
    name: str,
    namespace: str = "default",
    model_id: str = "meta-llama/Llama-3.1-8B-Instruct",
    num_shard: int = 1,
    image: str = TGI_ROCM_IMAGE,
    max_input_length: int = 4096,
    max_total_tokens: int = 8192,
    max_batch_prefill_tokens: int = 8192,
    extra_env: Optional[Dict[str, Any]] = None,
    cpu: str = "8",
    memory: str = "64Gi",
    annotations: Optional[Dict[str, str]] = None,
) -> Dict[str, Any]:
    env = {
        "MODEL_ID": model_id,
        "NUM_SHARD": num_shard,
        "MAX_CONCURRENT_REQUESTS": 256,
        "MAX_INPUT_LENGTH": max_input_length,
        "MAX_TOTAL_TOKENS": max_total_tokens,
        "MAX_BATCH_PREFILL_TOKENS": max_batch_prefill_tokens,
        "ROCM_USE_FLASH_ATTN_V2_TRITON": "false",
        "HSA_ENABLE_IPC_MODE_LEGACY": "0",
    }
    env.update(extra_env or {})
    container = {
        "name": "kserve-container",
        "image": image,
        "command": ["text-generation-launcher"],
        "args": ["--hostname", "0.0.0.0", "--port", "3000"],
        "env": _env(env),
        "ports": [{"containerPort": 3000, "protocol": "TCP"}],
        "resources": _resources(num_shard, cpu, memory),
        "volumeMounts": [{"name": "dshm", "mountPath": "/dev/shm"}],
    }
    predictor = {"containers": [container], "volumes": [{"name": "dshm", "emptyDir": {"medium": "Memory", "sizeLimit": "16Gi"}}]}
    return {"apiVersion": "serving.kserve.io/v1beta1", "kind": "InferenceService", "metadata": _meta(name, namespace, "tgi-rocm", annotations), "spec": {"predictor": predictor}}


def native_isvc(name: str, namespace: str = "default", model: str = "llama-3-8b", image: str = NATIVE_IMAGE, gpus: int = 1, serve_args: Optional[List[str]] = None, cpu: str = "8", memory: str = "64Gi", annotations: Optional[Dict[str, str]] = None) -> Dict[str, Any]:
    container = {
        "name": "kserve-container",
        "image": image,
        "command": ["kvmini", "serve"],
        "args": ["--host", "0.0.0.0", "--port", "8080", "--model", model, "--random-init"] + list(serve_args or []),
        "env": _env({"HSA_ENABLE_IPC_MODE_LEGACY": "0"}),
        "ports": [{"containerPort": 8080, "protocol": "TCP"}],
        "resources": _resources(gpus, cpu, memory),
    }
    return {"apiVersion": "serving.kserve.io/v1beta1", "kind": "InferenceService", "metadata": _meta(name, namespace, "native", annotations), "spec": {"predictor": {"containers": [container]}}}


def _workload(isvc: Dict[str, Any]) -> Dict[str, Any]:
    """The dict that carries ``resources`` (model predictor or first container)."""
    pred = isvc["spec"]["predictor"]
    if "model" in pred:
        return pred["model"]
    return pred["containers"][0]


def set_gpu_limit(isvc: Dict[str, Any], gpus: int, gpu_key: str = GPU_RESOURCE) -> Dict[str, Any]:
    out = copy.deepcopy(isvc)
    res = _workload(out).setdefault("resources", {"limits": {}, "requests": {}})
    for scope in ("limits", "requests"):
        d = res.setdefault(scope, {})
        for k in [k for k in d if k.startswith("amd.com/")]:
            del d[k]
        if gpus > 0:
            d[gpu_key] = str(gpus)
    return out


def apply_partition_profile(isvc: Dict[str, Any], profile: Dict[str, Any]) -> Dict[str, Any]:
    """Schedule the predictor onto a GPU partition.  Accepts both profile schemas:
    ``{resources: {limits, requests}, nodeSelector, tolerations, affinity}`` and
    ``{limits: {resource_key, count}, nodeSelector}``."""
    out = copy.deepcopy(isvc)
    wl = _workload(out)
    res = wl.setdefault("resources", {"limits": {}, "requests": {}})
    for scope in ("limits", "requests"):
        d = res.setdefault(scope, {})
        for k in [k for k in d if k.startswith("amd.com/")]:
            del d[k]
    if "resources" in profile:
        for scope in ("limits", "requests"):
            res[scope].update({k: str(v) for k, v in ((profile["resources"] or {}).get(scope) or {}).items()})
    elif "limits" in profile:
        lim = profile["limits"] or {}
        key, count = lim.get("resource_key", GPU_RESOURCE), lim.get("count", 1)
        res["limits"][key] = str(count)
        res["requests"][key] = str(count)
    pred = out["spec"]["predictor"]
    if profile.get("nodeSelector"):
        pred["nodeSelector"] = dict(profile["nodeSelector"])
    if profile.get("tolerations"):
        pred["tolerations"] = list(profile["tolerations"])
    if profile.get("affinity"):
        pred["affinity"] = dict(profile["affinity"])
    out["metadata"].setdefault("labels", {})["kvmini/partition"] = dns1123_label(str(profile.get("profile", "custom")))
    return out


def apply_autoscaling(isvc: Dict[str, Any], target: Optional[int] = None, initial_scale: Optional[int] = None, min_scale: Optional[int] = None, max_scale: Optional[int] = None, scale_to_zero_grace: Optional[str] = None, stable_window: Optional[str] = None, panic_window_pct: Optional[str] = None) -> Dict[str, Any]:
    out = copy.deepcopy(isvc)
    ann = out["metadata"].setdefault("annotations", {})
    for key, val in (
        ("autoscaling.knative.dev/target", target),
        ("autoscaling.knative.dev/initialScale", initial_scale),
        ("autoscaling.knative.dev/minScale", min_scale),
        ("autoscaling.knative.dev/maxScale", max_scale),
        ("autoscaling.knative.dev/scaleToZeroGracePeriod", scale_to_zero_grace),
        ("autoscaling.knative.dev/window", stable_window),
        ("autoscaling.knative.dev/panicWindowPercentage", panic_window_pct),
    ):
        if val is not None:
            ann[key] = str(val)
    return out


def to_yaml(manifest: Dict[str, Any]) -> str:
    import yaml

    return yaml.safe_dump(manifest, sort_keys=False, default_flow_style=False)


BUILDERS = {"vllm-rocm": vllm_rocm_isvc, "vllm": vllm_rocm_isvc, "tgi-rocm": tgi_rocm_isvc, "tgi": tgi_rocm_isvc, "native": native_isvc}
