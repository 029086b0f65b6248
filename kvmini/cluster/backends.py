"""Cluster backend implementations (kubectl / local process registry / fake).

All three return Kubernetes-shaped JSON (``items[].status.containerStatuses[]
.state.running.startedAt`` and friends) so the analyzer and the cost estimator
have exactly one code path.  The InferenceService label selector is the one
the reference uses: ``serving.kserve.io/inferenceservice=<name>``
(``analyze.py:371``, ``cost_estimator.py:166``).
"""

from __future__ import annotations

import datetime as dt
import json
import os
import shutil
import subprocess
import time
from pathlib import Path
from typing import Any, Dict, List, Optional

ISVC_LABEL = "serving.kserve.io/inferenceservice"
GPU_RESOURCE = "amd.com/gpu"
# CUs of an MI355X compute partition (8 XCDs x 32 CUs), by the name a partition profile carries
PARTITION_CUS = {"cpx": 32, "qpx": 64, "dpx": 128, "mi355x-cpx-nps4": 32, "mi355x-qpx-nps1": 64, "mi355x-dpx-nps1": 128}


def iso_utc(ts: float) -> str:
    return dt.datetime.fromtimestamp(ts, tz=dt.timezone.utc).strftime("%Y-%m-%dT%H:%M:%S.%fZ")


def parse_k8s_time(s: str) -> float:
    """RFC-3339 timestamp as written by the API server -> epoch seconds."""
    return dt.datetime.fromisoformat(s.replace("Z", "+00:00")).timestamp()


def make_pod(
    name: str,
    service: str,
    namespace: str = "default",
    started_at: Optional[float] = None,
    finished_at: Optional[float] = None,
    node: str = "node-0",
    gpus: float = 1,
    cpu: str = "4",
    memory: str = "16Gi",
    image: str = "rocm/vllm:latest",
    gpu_resource: str = GPU_RESOURCE,
    sidecar: bool = False,
) -> Dict[str, Any]:
    """Build a pod object of the shape ``kubectl get pods -o json`` returns."""
    started_at = time.time() if started_at is None else started_at
    state: Dict[str, Any]
    if finished_at is None:
        state = {"running": {"startedAt": iso_utc(started_at)}}
    else:
        state = {"terminated": {"startedAt": iso_utc(started_at), "finishedAt": iso_utc(finished_at)}}
    res = {"cpu": cpu, "memory": memory}
    if gpus:
        res[gpu_resource] = str(gpus) if float(gpus) != int(gpus) else str(int(gpus))
    containers = [{"name": "kserve-container", "image": image, "resources": {"requests": dict(res), "limits": dict(res)}}]
    statuses = [{"name": "kserve-container", "image": image, "imageID": f"{image}@sha256:" + "0" * 64, "state": state, "ready": finished_at is None}]
    if sidecar:
        sc_res = {"cpu": "100m", "memory": "128Mi"}
        containers.append({"name": "queue-proxy", "image": "knative/queue:latest", "resources": {"requests": sc_res, "limits": sc_res}})
        statuses.append({"name": "queue-proxy", "image": "knative/queue:latest", "state": state, "ready": True})
    return {
        "apiVersion": "v1",
        "kind": "Pod",
        "metadata": {"name": name, "namespace": namespace, "labels": {ISVC_LABEL: service, "app": service}},
        "spec": {"nodeName": node, "containers": containers},
        "status": {"phase": "Running" if finished_at is None else "Succeeded", "startTime": iso_utc(started_at), "containerStatuses": statuses},
    }


class ClusterBackend:
    """Interface every stage programs against.  All methods degrade to empty
    results instead of raising when the cluster cannot answer (the reference's
    analyzer does that, its cost estimator does not: ``cost_estimator.py:426``)."""

    name = "abstract"

    def available(self) -> bool:
        return True

    # --- read side -------------------------------------------------------
    def get_pods(self, namespace: str, service: str) -> List[Dict[str, Any]]:
        return []

    def get_node(self, node_name: str) -> Dict[str, Any]:
        return {}

    def get_isvc(self, namespace: str, service: str) -> Dict[str, Any]:
        return {}

    def get_isvc_url(self, namespace: str, service: str) -> Optional[str]:
        return (self.get_isvc(namespace, service).get("status") or {}).get("url")

    def logs(self, namespace: str, service: str, tail: int = 200) -> str:
        return ""

    def cluster_facts(self) -> Dict[str, Any]:
        return {"backend": self.name}

    def usage_snapshot(self, namespace: str, service: str) -> Optional[Dict[str, Any]]:
        """Measured resource use of the service's processes so far: ``{"t": wall clock,
        "cpu_seconds": user + system time of every process, "rss_bytes": resident memory now,
        "peak_rss_bytes": its high-water mark, "processes": n}``; None when this backend cannot
        measure (the cost record then prices the declared requests / limits)."""
        return None

    # --- write side ------------------------------------------------------
    def apply(self, manifest: Dict[str, Any]) -> None:
        raise NotImplementedError

    def delete_isvc(self, namespace: str, service: str) -> None:
        raise NotImplementedError

    def wait_ready(self, namespace: str, service: str, timeout_s: float = 600.0) -> float:
        """Block until the service is Ready; returns seconds waited."""
        raise NotImplementedError

    # --- fault injection (chaos harness) ---------------------------------
    def inject_fault(self, fault: str, namespace: str, service: str) -> Dict[str, Any]:
        raise NotImplementedError(f"backend {self.name} cannot inject {fault}")

    def clear_fault(self, fault: str, namespace: str, service: str) -> None:
        """Undo a fault that persists until removed (network shaping); no-op otherwise."""
        return None

    # --- derived ---------------------------------------------------------
    def cold_start_times(self, namespace: str, service: str, start: float, end: float) -> List[float]:
        """Epoch seconds at which a container of the service started inside
        ``[start, end]`` — one per pod (reference ``analyze.py:358-395``)."""
        out: List[float] = []
        for pod in self.get_pods(namespace, service):
            for cs in (pod.get("status") or {}).get("containerStatuses", []) or []:
                started = ((cs.get("state") or {}).get("running") or {}).get("startedAt")
                if not started:
                    continue
                try:
                    ts = parse_k8s_time(started)
                except ValueError:
                    continue
                if start <= ts <= end:
                    out.append(ts)
                    break
        return sorted(out)


class KubectlBackend(ClusterBackend):
    name = "kubectl"

    def __init__(self, kubectl: str = "kubectl", context: Optional[str] = None):
        self.kubectl = kubectl
        self.context = context

    def available(self) -> bool:
        return shutil.which(self.kubectl) is not None

    def _run(self, args: List[str], input_text: Optional[str] = None, timeout: float = 60.0) -> str:
        cmd = [self.kubectl]
        if self.context:
            cmd += ["--context", self.context]
        cmd += args
        return subprocess.run(cmd, input=input_text, capture_output=True, text=True, check=True, timeout=timeout).stdout

    def _json(self, args: List[str]) -> Dict[str, Any]:
        try:
            return json.loads(self._run(args + ["-o", "json"]))
        except (subprocess.SubprocessError, OSError, ValueError):
            return {}

    def get_pods(self, namespace: str, service: str) -> List[Dict[str, Any]]:
        return self._json(["get", "pods", "-n", namespace, "-l", f"{ISVC_LABEL}={service}"]).get("items", [])

    def get_node(self, node_name: str) -> Dict[str, Any]:
        return self._json(["get", "node", node_name])

    def get_isvc(self, namespace: str, service: str) -> Dict[str, Any]:
        return self._json(["get", "inferenceservice", service, "-n", namespace])

    def logs(self, namespace: str, service: str, tail: int = 200) -> str:
        try:
            return self._run(["-n", namespace, "logs", "-l", f"{ISVC_LABEL}={service}", f"--tail={tail}"])
        except (subprocess.SubprocessError, OSError):
            return ""

    def apply(self, manifest: Dict[str, Any]) -> None:
        ns = (manifest.get("metadata") or {}).get("namespace")
        if ns:
            try:
                self._run(["create", "namespace", ns])
            except subprocess.SubprocessError:
                pass  # already exists
        self._run(["apply", "-f", "-"], input_text=json.dumps(manifest))

    def delete_isvc(self, namespace: str, service: str) -> None:
        try:
            self._run(["delete", "inferenceservice", service, "-n", namespace, "--ignore-not-found"])
        except subprocess.SubprocessError:
            pass

    def wait_ready(self, namespace: str, service: str, timeout_s: float = 600.0) -> float:
        t0 = time.time()
        self._run(
            ["wait", "--for=condition=Ready", f"inferenceservice/{service}", "-n", namespace, f"--timeout={int(timeout_s)}s"],
            timeout=timeout_s + 30,
        )
        return time.time() - t0

    def cluster_facts(self) -> Dict[str, Any]:
        facts: Dict[str, Any] = {"backend": self.name}
        ver = self._json(["version"])
        facts["kubernetes_version"] = (ver.get("serverVersion") or {}).get("gitVersion")
        nodes = self._json(["get", "nodes"]).get("items", [])
        gpu_nodes = []
        for n in nodes:
            cap = (n.get("status") or {}).get("capacity") or {}
            labels = (n.get("metadata") or {}).get("labels") or {}
            if GPU_RESOURCE in cap:
                gpu_nodes.append(
                    {
                        "name": n["metadata"]["name"],
                        "gpu_count": cap.get(GPU_RESOURCE),
                        "gpu_product": labels.get("amd.com/gpu.product-name") or labels.get("amd.com/gpu.family"),
                        "gpu_vram": labels.get("amd.com/gpu.vram"),
                        "compute_partition": labels.get("amd.com/compute-partitioning-supported"),
                        "instance_type": labels.get("node.kubernetes.io/instance-type"),
                        "zone": labels.get("topology.kubernetes.io/zone"),
                    }
                )
        facts["gpu_nodes"] = gpu_nodes
        return facts

    def inject_fault(self, fault: str, namespace: str, service: str) -> Dict[str, Any]:
        pods = self.get_pods(namespace, service)
        pod = pods[0]["metadata"]["name"] if pods else None
        if fault == "device-plugin-restart":
            # AMD GPU operator / k8s-device-plugin daemonset (amdgpu), any namespace
            ds = self._json(["get", "daemonsets", "-A"]).get("items", [])
            hit = [d for d in ds if "amdgpu" in d["metadata"]["name"] or "amd-gpu" in d["metadata"]["name"] or "device-plugin" in d["metadata"]["name"]]
            for d in hit[:1]:
                self._run(["-n", d["metadata"]["namespace"], "rollout", "restart", f"daemonset/{d['metadata']['name']}"])
            return {"fault": fault, "target": [d["metadata"]["name"] for d in hit[:1]]}
        if pod is None:
            raise RuntimeError("no predictor pod to inject a fault into")
        if fault == "pod-delete":
            self._run(["-n", namespace, "delete", "pod", pod, "--grace-period=0", "--force"])
        elif fault == "container-kill":
            self._run(["-n", namespace, "exec", pod, "--", "sh", "-c", "kill -9 1"])
        elif fault == "network-degrade":
            self._run(["-n", namespace, "exec", pod, "--", "sh", "-c", "tc qdisc add dev eth0 root netem loss 10% delay 50ms"])
        elif fault == "node-drain":
            node = pods[0]["spec"].get("nodeName")
            self._run(["drain", node, "--ignore-daemonsets", "--delete-emptydir-data", "--force", "--timeout=120s"], timeout=180)
            self._run(["uncordon", node])
        else:
            raise ValueError(f"unknown fault {fault}")
        return {"fault": fault, "target": pod}

    def clear_fault(self, fault: str, namespace: str, service: str) -> None:
        if fault != "network-degrade":
            return
        pods = self.get_pods(namespace, service)
        if pods:
            try:
                self._run(["-n", namespace, "exec", pods[0]["metadata"]["name"], "--", "sh", "-c", "tc qdisc del dev eth0 root"])
            except (subprocess.SubprocessError, OSError):
                pass


class FakeBackend(ClusterBackend):
    """In-memory cluster.  ``state`` layout::

        {"pods": [<pod json>...], "nodes": {<name>: <node json>}, "isvc": {"<ns>/<name>": {...}}}

    ``KVMINI_FAKE_CLUSTER=<file.json>`` preloads it (used by CLI-level tests)."""

    name = "fake"

    def __init__(self, state: Optional[Dict[str, Any]] = None, ready_delay_s: float = 0.0):
        self.state: Dict[str, Any] = {"pods": [], "nodes": {}, "isvc": {}, "logs": ""}
        if state:
            self.state.update(state)
        self.ready_delay_s = ready_delay_s
        self.applied: List[Dict[str, Any]] = []
        self.faults: List[str] = []

    @classmethod
    def from_file(cls, path: str) -> "FakeBackend":
        with open(path) as f:
            return cls(json.load(f))

    def add_pod(self, **kw: Any) -> Dict[str, Any]:
        pod = make_pod(**kw)
        self.state["pods"].append(pod)
        return pod

    def add_node(self, name: str = "node-0", product: str = "AMD_Instinct_MI355X", gpus: int = 8, extra_labels: Optional[Dict[str, str]] = None) -> Dict[str, Any]:
        labels = {
            "amd.com/gpu.product-name": product,
            "amd.com/gpu.family": "AI",
            "amd.com/gpu.vram": "288G",
            "amd.com/gpu.cu-count": "256",
            "kubernetes.io/hostname": name,
        }
        labels.update(extra_labels or {})
        node = {
            "apiVersion": "v1",
            "kind": "Node",
            "metadata": {"name": name, "labels": labels},
            "status": {"capacity": {GPU_RESOURCE: str(gpus), "cpu": "128", "memory": "2048Gi"}},
        }
        self.state["nodes"][name] = node
        return node

    def get_pods(self, namespace: str, service: str) -> List[Dict[str, Any]]:
        out = []
        for p in self.state["pods"]:
            md = p.get("metadata") or {}
            if (md.get("labels") or {}).get(ISVC_LABEL) == service and md.get("namespace", namespace) == namespace:
                out.append(p)
        return out

    def get_node(self, node_name: str) -> Dict[str, Any]:
        return self.state["nodes"].get(node_name, {})

    def get_isvc(self, namespace: str, service: str) -> Dict[str, Any]:
        return self.state["isvc"].get(f"{namespace}/{service}", {})

    def logs(self, namespace: str, service: str, tail: int = 200) -> str:
        return self.state.get("logs", "")

    def apply(self, manifest: Dict[str, Any]) -> None:
        self.applied.append(manifest)
        md = manifest.get("metadata") or {}
        if manifest.get("kind") == "InferenceService":
            ns, name = md.get("namespace", "default"), md.get("name")
            m = dict(manifest)
            m["status"] = {"url": f"http://{name}.{ns}.example.com", "conditions": [{"type": "Ready", "status": "True"}]}
            self.state["isvc"][f"{ns}/{name}"] = m
            if not self.get_pods(ns, name):
                self.add_pod(name=f"{name}-predictor-00001-deployment-abcde", service=name, namespace=ns)

    def delete_isvc(self, namespace: str, service: str) -> None:
        self.state["isvc"].pop(f"{namespace}/{service}", None)
        self.state["pods"] = [p for p in self.state["pods"] if not ((p["metadata"].get("labels") or {}).get(ISVC_LABEL) == service and p["metadata"].get("namespace") == namespace)]

    def wait_ready(self, namespace: str, service: str, timeout_s: float = 600.0) -> float:
        if self.ready_delay_s:
            time.sleep(min(self.ready_delay_s, timeout_s))
        return self.ready_delay_s

    def inject_fault(self, fault: str, namespace: str, service: str) -> Dict[str, Any]:
        self.faults.append(fault)
        # a restarted pod shows up as a fresh container start
        if fault in ("pod-delete", "container-kill", "node-drain"):
            for p in self.get_pods(namespace, service):
                for cs in p["status"]["containerStatuses"]:
                    cs["state"] = {"running": {"startedAt": iso_utc(time.time())}}
        return {"fault": fault, "target": "fake"}

    def cluster_facts(self) -> Dict[str, Any]:
        return {"backend": self.name, "gpu_nodes": [{"name": n, "gpu_count": v["status"]["capacity"].get(GPU_RESOURCE), "gpu_product": v["metadata"]["labels"].get("amd.com/gpu.product-name")} for n, v in self.state["nodes"].items()]}


class LocalBackend(ClusterBackend):
    """Servers running on this host.  ``kvmini serve`` / ``kvmini mock`` register
    themselves in a small JSON registry (one record per server process: pid, port,
    start time, ready time, GPU count, restarts).  The registry is what makes the
    cold/warm split and the resource-seconds cost model work without Kubernetes:
    a (re)start of the server process inside the measurement window is a cold
    start; its resources are ``n_gpus`` x ``amd.com/gpu`` plus the host CPU/memory
    share it was started with."""

    name = "local"

    def __init__(self, registry: Optional[str] = None):
        self.registry = Path(registry or os.environ.get("KVMINI_LOCAL_REGISTRY", "runs/.local_cluster.json"))

    # registry -----------------------------------------------------------
    def _load(self) -> Dict[str, Any]:
        try:
            with open(self.registry) as f:
                return json.load(f)
        except (OSError, ValueError):
            return {"servers": []}

    def _save(self, reg: Dict[str, Any]) -> None:
        self.registry.parent.mkdir(parents=True, exist_ok=True)
        tmp = self.registry.with_suffix(".tmp%d" % os.getpid())
        with open(tmp, "w") as f:
            json.dump(reg, f, indent=2)
        os.replace(tmp, self.registry)

    def register_server(
        self,
        service: str,
        url: str,
        pid: int,
        started_at: float,
        namespace: str = "local",
        gpus: float = 1,
        cpu: str = "8",
        memory: str = "64Gi",
        image: str = "kvmini-native",
        gpu_product: Optional[str] = None,
    ) -> None:
        reg = self._load()
        reg["servers"].append(
            {
                "service": service,
                "namespace": namespace,
                "url": url,
                "pid": pid,
                "started_at": started_at,
                "finished_at": None,
                "gpus": gpus,
                "cpu": cpu,
                "memory": memory,
                "image": image,
                "gpu_product": gpu_product,
            }
        )
        self._save(reg)

    def mark_finished(self, pid: int, finished_at: Optional[float] = None) -> None:
        reg = self._load()
        for s in reg["servers"]:
            if s["pid"] == pid and s.get("finished_at") is None:
                s["finished_at"] = finished_at or time.time()
        self._save(reg)

    # read side ------------------------------------------------------------
    def _records(self, namespace: str, service: str) -> List[Dict[str, Any]]:
        return [s for s in self._load()["servers"] if s["service"] == service and s.get("namespace", namespace) in (namespace, "local")]

    def get_pods(self, namespace: str, service: str) -> List[Dict[str, Any]]:
        pods = []
        dep = (self._load().get("isvc") or {}).get(f"{namespace}/{service}") or {}
        for i, s in enumerate(self._records(namespace, service)):
            pods.append(
                make_pod(
                    name=f"{service}-predictor-local-{s['pid']}-{i}",
                    service=service,
                    namespace=namespace,
                    started_at=s["started_at"],
                    finished_at=s.get("finished_at"),
                    node="localhost",
                    gpus=s.get("gpus", 1),
                    cpu=str(s.get("cpu", "8")),
                    memory=str(s.get("memory", "64Gi")),
                    image=s.get("image", "kvmini-native"),
                    gpu_resource=s.get("gpu_resource") or dep.get("gpu_resource") or GPU_RESOURCE,
                )
            )
        return pods

    @staticmethod
    def _process_tree(roots: List[int]) -> List[int]:
        """``roots`` and all their descendants that are alive (one pass over /proc)."""
        parent: Dict[int, int] = {}
        for name in os.listdir("/proc"):
            if not name.isdigit():
                continue
            try:
                with open(f"/proc/{name}/stat", "rb") as f:
                    rest = f.read().rsplit(b")", 1)[1].split()
                parent[int(name)] = int(rest[1])  # field 4 (ppid); the split starts at field 3
            except (OSError, IndexError, ValueError):
                continue
        want = {r for r in roots if r in parent}
        grew = True
        while grew:
            grew = False
            for pid, pp in parent.items():
                if pp in want and pid not in want:
                    want.add(pid)
                    grew = True
        return sorted(want)

    def usage_snapshot(self, namespace: str, service: str) -> Optional[Dict[str, Any]]:
        """user + system CPU seconds and resident memory of the service's process trees (the
        server, its tensor-parallel ranks, a router's replicas), read from ``/proc/<pid>/stat``
        and ``/proc/<pid>/status``.  Two snapshots around a load test give the CPU-seconds the
        test really used; the peak RSS is what the memory price is charged on."""
        reg = self._load()
        roots = [int(s["pid"]) for s in self._records(namespace, service) if s.get("finished_at") is None]
        dep = (reg.get("isvc") or {}).get(f"{namespace}/{service}") or {}
        if dep.get("pid"):
            roots.append(int(dep["pid"]))
        pids = self._process_tree(sorted(set(roots)))
        if not pids:
            return None
        tick = float(os.sysconf("SC_CLK_TCK"))
        cpu = rss = peak = 0.0
        n = 0
        for pid in pids:
            try:
                with open(f"/proc/{pid}/stat", "rb") as f:
                    rest = f.read().rsplit(b")", 1)[1].split()
                with open(f"/proc/{pid}/status") as f:
                    status = dict(line.split(":", 1) for line in f if ":" in line)
            except OSError:
                continue
            cpu += (int(rest[11]) + int(rest[12])) / tick  # utime + stime
            rss += float(status.get("VmRSS", "0 kB").split()[0]) * 1024.0
            peak += float(status.get("VmHWM", "0 kB").split()[0]) * 1024.0
            n += 1
        if n == 0:
            return None
        return {"t": time.time(), "cpu_seconds": cpu, "rss_bytes": rss, "peak_rss_bytes": peak, "processes": n}

    def get_node(self, node_name: str) -> Dict[str, Any]:
        product = None
        for s in self._load()["servers"]:
            product = s.get("gpu_product") or product
        if product is None:
            try:
                from kvmini.telemetry.amdsmi_sampler import gpu_inventory

                inv = gpu_inventory()
                product = inv[0]["product"] if inv else None
            except Exception:
                product = None
        labels = {"kubernetes.io/hostname": node_name}
        if product:
            labels["amd.com/gpu.product-name"] = product.replace(" ", "_")
        return {"kind": "Node", "metadata": {"name": node_name, "labels": labels}, "status": {"capacity": {}}}

    def get_isvc(self, namespace: str, service: str) -> Dict[str, Any]:
        dep = (self._load().get("isvc") or {}).get(f"{namespace}/{service}")
        if dep and self._alive(int(dep.get("pid", 0))):
            m = dict(dep.get("manifest") or {"kind": "InferenceService", "metadata": {"name": service, "namespace": namespace}})
            m["status"] = {"url": dep["url"], "replicas": dep.get("replicas", 1), "gpus_per_replica": dep.get("gpus_per_replica", 1)}
            return m
        recs = [s for s in self._records(namespace, service) if s.get("finished_at") is None]
        if not recs:
            return {}
        return {"kind": "InferenceService", "metadata": {"name": service, "namespace": namespace}, "status": {"url": recs[-1]["url"]}}

    # write side -------------------------------------------------------------
    @staticmethod
    def _alive(pid: int) -> bool:
        if pid <= 0:
            return False
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        except PermissionError:
            return True
        try:  # a child of this process that has exited but was not waited for still answers signal 0
            with open(f"/proc/{pid}/stat") as f:
                return f.read().rsplit(")", 1)[1].split()[0] != "Z"
        except OSError:
            return True

    @staticmethod
    def local_spec(manifest: Dict[str, Any]) -> Dict[str, Any]:
        """What an InferenceService manifest means on a bare node: which server to start with which
        arguments, how many replicas, how many GPUs each.

        * the native engine's manifests (``kvmini.deploy.manifests.native_isvc``: a container whose command
          is ``kvmini serve``) run as they are: the container's args minus the in-cluster host / port;
        * replicas: ``autoscaling.knative.dev/initialScale``, else ``minScale``, at least one (scale to
          zero has no meaning for a process that is started on demand);
        * GPUs per replica: the ``amd.com/*`` limit of the container (a whole GPU or a partition);
          ``--tensor-parallel-size`` follows it unless the args already say;
        * other runtimes' manifests (vLLM-ROCm, TGI-ROCm) cannot run here; with
          ``KVMINI_LOCAL_EMULATE=mock`` they are stood in for by the protocol mock, else this raises."""
        md = manifest.get("metadata") or {}
        pred = ((manifest.get("spec") or {}).get("predictor")) or {}
        wl = pred.get("model") or ((pred.get("containers") or [{}])[0])
        limits = ((wl.get("resources") or {}).get("limits")) or {}
        gpus, gpu_key = 0, GPU_RESOURCE
        for k, v in limits.items():
            if k.startswith("amd.com/"):
                try:
                    gpus, gpu_key = max(gpus, int(float(v))), k
                except (TypeError, ValueError):
                    pass
        ann = md.get("annotations") or {}
        replicas = 1
        for key in ("autoscaling.knative.dev/initialScale", "autoscaling.knative.dev/minScale"):
            if str(ann.get(key, "")).isdigit():
                replicas = max(1, int(ann[key]))
                break
        cmd = list(wl.get("command") or [])
        args = [str(x) for x in (wl.get("args") or [])]
        kind = "native" if cmd[:2] == ["kvmini", "serve"] else "mock" if cmd[:2] == ["kvmini", "mock"] else None
        if kind is None:
            if os.environ.get("KVMINI_LOCAL_EMULATE") == "mock":
                kind, args = "mock", []
            else:
                raise ValueError(f"the local backend runs this repository's servers only (manifest command {cmd or (wl.get('runtime') or wl.get('image'))!r}); "
                                 "deploy --backend native, or set KVMINI_LOCAL_EMULATE=mock to stand the runtime in with the protocol mock")
        keep: List[str] = []
        i = 0
        while i < len(args):  # the host / port of the in-cluster container are this backend's to choose
            if args[i] in ("--host", "--port"):
                i += 2
                continue
            keep.append(args[i])
            i += 1
        return {"kind": kind, "name": md.get("name", "demo-llm"), "namespace": md.get("namespace", "local"), "args": keep, "replicas": replicas, "gpus": gpus,
                "partition": (md.get("labels") or {}).get("kvmini/partition"), "gpu_resource": gpu_key, "env": {e["name"]: str(e.get("value", "")) for e in (wl.get("env") or []) if "name" in e}}

    def apply(self, manifest: Dict[str, Any]) -> None:
        """Start what the manifest describes as fresh child processes of this one: ``kvmini serve`` for
        one replica (which starts its own tensor-parallel ranks), ``kvmini router --replicas N`` in
        front of N of them.  Applying a service that already runs replaces it, as ``kubectl apply`` of
        a changed spec rolls the revision (reference flow: ``deploy.sh:108-114``,
        ``sweeps/autoscale-sweep.sh:131-163``, ``scripts/compare_backends.py:244-345``).  Returns
        once the processes exist; :meth:`wait_ready` waits for ``/health``."""
        import socket
        import sys

        if manifest.get("kind") != "InferenceService":
            return  # namespaces, secrets, serving runtimes: nothing to do on a bare node
        spec = self.local_spec(manifest)
        ns, name = spec["namespace"], spec["name"]
        self.delete_isvc(ns, name)
        want_port = str(((manifest.get("metadata") or {}).get("annotations") or {}).get("kvmini/local-port", ""))
        if want_port.isdigit() and int(want_port) > 0:
            port = int(want_port)
        else:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = int(sk.getsockname()[1])
        args = list(spec["args"])
        cpu_only = "--device" in args and args[args.index("--device") + 1] == "cpu"
        no_gpu = os.environ.get("KVMINI_LOCAL_DEVICE") == "cpu"
        if spec["kind"] == "native" and no_gpu and not cpu_only:
            args += ["--device", "cpu"]
            cpu_only = True
        tp = max(1, spec["gpus"]) if spec["kind"] == "native" and not cpu_only else 1
        if "--tensor-parallel-size" in args:
            tp = int(args[args.index("--tensor-parallel-size") + 1])
        elif tp > 1:
            args += ["--tensor-parallel-size", str(tp)]
        py = [sys.executable, "-m", "kvmini"]
        if spec["kind"] == "mock":
            cmd = py + ["mock", "--port", str(port), "--service", name, "--register"] + args
        elif spec["replicas"] > 1:
            front = ["router", "--port", str(port), "--replicas", str(spec["replicas"]), "--gpus-per-replica", str(tp), "--service", name, "--namespace", ns]
            if cpu_only:
                front += ["--device", "cpu"]
            inner = [a for a in args]
            if "--tensor-parallel-size" in inner:  # the router hands the degree to every replica itself
                k = inner.index("--tensor-parallel-size")
                del inner[k : k + 2]
            cmd = py + front + ["--"] + inner
        else:
            cmd = py + ["serve", "--port", str(port), "--service", name, "--namespace", ns] + args
        log_dir = self.registry.parent / ".local_logs"
        log_dir.mkdir(parents=True, exist_ok=True)
        log = open(log_dir / f"{name}.log", "ab")
        pkg_root = str(Path(__file__).resolve().parents[2])
        env = dict(os.environ)
        env.update({k: v for k, v in spec["env"].items() if k.isupper()})
        env["PYTHONPATH"] = pkg_root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        env["KVMINI_LOCAL_REGISTRY"] = str(self.registry.resolve())
        # A compute partition cannot be switched without root on a bare node; what a partition profile
        # changes for the engine is the number of CUs it gets, and that the HSA runtime can restrict per
        # process (CU mask: CPX = one XCD = 32 CUs, QPX 64, DPX 128).  The whole HBM stays visible — the
        # cost model prices the partition's share through the resource key either way.
        cus = PARTITION_CUS.get(str(spec["partition"] or "").lower())
        if cus and not cpu_only and os.environ.get("KVMINI_LOCAL_PARTITION_EMULATION", "cu-mask") == "cu-mask":
            env["HSA_CU_MASK"] = f"0:0-{cus - 1}"
        proc = subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT, env=env, start_new_session=True)
        self._children = getattr(self, "_children", {})
        self._children[proc.pid] = proc
        reg = self._load()
        reg.setdefault("isvc", {})[f"{ns}/{name}"] = {"url": f"http://127.0.0.1:{port}", "pid": proc.pid, "applied_at": time.time(), "replicas": spec["replicas"], "gpus_per_replica": tp,
                                                      "kind": spec["kind"], "partition": spec["partition"], "gpu_resource": spec["gpu_resource"], "cu_mask": env.get("HSA_CU_MASK"), "command": cmd, "log": str(log_dir / f"{name}.log"), "manifest": manifest}
        self._save(reg)

    def wait_ready(self, namespace: str, service: str, timeout_s: float = 600.0) -> float:
        import urllib.request

        t0 = time.time()
        while time.time() - t0 < timeout_s:
            dep = (self._load().get("isvc") or {}).get(f"{namespace}/{service}")
            if dep:
                child = getattr(self, "_children", {}).get(int(dep["pid"]))
                if (child is not None and child.poll() is not None) or not self._alive(int(dep["pid"])):
                    raise RuntimeError(f"{service}: the server process exited before it became ready (see {dep.get('log')})")
            url = self.get_isvc_url(namespace, service)
            if url:
                try:
                    with urllib.request.urlopen(url.rstrip("/") + "/health", timeout=2) as r:
                        if r.status == 200:
                            return time.time() - t0
                except Exception:
                    pass
            time.sleep(0.2)
        raise TimeoutError(f"{service} not ready after {timeout_s}s")

    def delete_isvc(self, namespace: str, service: str) -> None:
        import signal

        reg = self._load()
        dep = (reg.get("isvc") or {}).pop(f"{namespace}/{service}", None)
        if dep:
            self._save(reg)
            pid = int(dep.get("pid", 0))
            if self._alive(pid):
                # the deployment was started in a session of its own: the signal reaches the router, the
                # replicas and the tensor-parallel ranks alike
                for sig, wait_s in ((signal.SIGTERM, 15.0), (signal.SIGKILL, 5.0)):
                    try:
                        os.killpg(pid, sig)
                    except OSError:
                        try:
                            os.kill(pid, sig)
                        except OSError:
                            break
                    child = getattr(self, "_children", {}).get(pid)
                    t_end = time.time() + wait_s
                    while time.time() < t_end:
                        if child is not None:
                            if child.poll() is not None:
                                break
                        elif not self._alive(pid):
                            break
                        time.sleep(0.05)
                    else:
                        continue
                    break
            getattr(self, "_children", {}).pop(pid, None)
        for s in self._records(namespace, service):
            if s.get("finished_at") is None:
                try:
                    os.kill(int(s["pid"]), signal.SIGTERM)
                except OSError:
                    pass
                self.mark_finished(s["pid"])

    def inject_fault(self, fault: str, namespace: str, service: str) -> Dict[str, Any]:
        """Local analogues of the five cluster faults: kill / stall the server
        process; the supervisor (``kvmini serve --supervise``) restarts it."""
        import signal

        live = [s for s in self._records(namespace, service) if s.get("finished_at") is None]
        if not live:
            raise RuntimeError("no live local server")
        pid = int(live[-1]["pid"])
        if fault in ("pod-delete", "container-kill"):
            os.kill(pid, signal.SIGKILL)
            self.mark_finished(pid)
        elif fault in ("network-degrade", "device-plugin-restart", "node-drain"):
            # stall: stop the process for 2 s, then continue
            os.kill(pid, signal.SIGSTOP)
            time.sleep(2.0)
            os.kill(pid, signal.SIGCONT)
        else:
            raise ValueError(f"unknown fault {fault}")
        return {"fault": fault, "target": pid}

    def cluster_facts(self) -> Dict[str, Any]:
        facts: Dict[str, Any] = {"backend": self.name, "servers": self._load()["servers"]}
        try:
            from kvmini.telemetry.amdsmi_sampler import gpu_inventory

            facts["gpus"] = gpu_inventory()
        except Exception:
            facts["gpus"] = []
        return facts


def get_backend(name: Optional[str] = None) -> ClusterBackend:
    """Resolve the backend: explicit name > ``KVMINI_CLUSTER_BACKEND`` > auto
    (kubectl if installed, else the local registry if it exists, else an empty fake)."""
    name = name or os.environ.get("KVMINI_CLUSTER_BACKEND", "auto")
    if name == "kubectl":
        return KubectlBackend()
    if name == "local":
        return LocalBackend()
    if name == "fake":
        path = os.environ.get("KVMINI_FAKE_CLUSTER")
        return FakeBackend.from_file(path) if path and os.path.exists(path) else FakeBackend()
    if name != "auto":
        raise ValueError(f"unknown cluster backend {name!r}")
    fake_path = os.environ.get("KVMINI_FAKE_CLUSTER")
    if fake_path and os.path.exists(fake_path):
        return FakeBackend.from_file(fake_path)
    kb = KubectlBackend()
    if kb.available():
        return kb
    lb = LocalBackend()
    if lb.registry.exists():
        return lb
    return FakeBackend()
