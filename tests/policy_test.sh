#!/usr/bin/env bash
# End-to-end check of the admission policies (policies/kyverno, policies/gatekeeper) against a
# real or kind cluster.  Needs kubectl with cluster-admin; installs nothing unless --install is
# given.  Writes policy-test-report.md.  Not part of the pytest suite (no cluster in CI boxes).
#
#   tests/policy_test.sh [--engine gatekeeper|kyverno] [--install] [--namespace policy-test] [--keep]
set -euo pipefail

ENGINE=gatekeeper; INSTALL=0; NS=policy-test; KEEP=0
while [[ $# -gt 0 ]]; do
  case "$1" in
    --engine) ENGINE="$2"; shift 2 ;;
    --install) INSTALL=1; shift ;;
    --namespace) NS="$2"; shift 2 ;;
    --keep) KEEP=1; shift ;;
    -h|--help) sed -n 2,8p "$0"; exit 0 ;;
    *) echo "unknown flag $1" >&2; exit 2 ;;
  esac
done
command -v kubectl >/dev/null 2>&1 || { echo "kubectl is required" >&2; exit 2; }
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REPORT="$ROOT/policy-test-report.md"
pass=0; fail=0
note() { echo "$*" | tee -a "$REPORT"; }
check() {  # name expected(violation|clean) actual
  if [[ "$2" == "$3" ]]; then pass=$((pass+1)); note "- PASS  $1 ($3)"; else fail=$((fail+1)); note "- FAIL  $1 (expected $2, got $3)"; fi
}

: > "$REPORT"
note "# Policy test report ($(date -u +%FT%TZ), engine: $ENGINE)"

if [[ $INSTALL -eq 1 ]]; then
  if [[ "$ENGINE" == gatekeeper ]]; then
    kubectl apply -f https://raw.githubusercontent.com/open-policy-agent/gatekeeper/v3.15.0/deploy/gatekeeper.yaml
    kubectl -n gatekeeper-system rollout status deploy/gatekeeper-controller-manager --timeout=180s
  else
    kubectl create -f https://github.com/kyverno/kyverno/releases/download/v1.12.0/install.yaml
    kubectl -n kyverno rollout status deploy/kyverno-admission-controller --timeout=180s
  fi
fi

kubectl get ns "$NS" >/dev/null 2>&1 || kubectl create ns "$NS"
if [[ "$ENGINE" == gatekeeper ]]; then
  kubectl apply -f "$ROOT/policies/gatekeeper/constrainttemplates.yaml"
  sleep 5   # CRDs of the templates must exist before the constraints
  kubectl apply -f "$ROOT/policies/gatekeeper/constraints.yaml"
else
  kubectl apply -f "$ROOT/policies/kyverno/"
fi

pod() {  # name, extra container securityContext yaml, extra pod-spec yaml, resources yaml
  cat <<YAML
apiVersion: v1
kind: Pod
metadata: {name: $1, namespace: $NS, labels: {app: policy-test}}
spec:
  restartPolicy: Never
$3
  containers:
  - name: c
    image: registry.k8s.io/pause:3.9
    securityContext: $2
    resources: $4
YAML
}
OK_SC='{runAsNonRoot: true, runAsUser: 65534, readOnlyRootFilesystem: true, allowPrivilegeEscalation: false}'
OK_RES='{requests: {cpu: 10m, memory: 16Mi}, limits: {cpu: 10m, memory: 16Mi}}'

note ""; note "## Test pods"
pod compliant       "$OK_SC" "" "$OK_RES" | kubectl apply -f - >/dev/null
pod runs-as-root    '{runAsUser: 0, readOnlyRootFilesystem: true}' "" "$OK_RES" | kubectl apply -f - >/dev/null || true
pod writable-rootfs '{runAsNonRoot: true, runAsUser: 65534, readOnlyRootFilesystem: false}' "" "$OK_RES" | kubectl apply -f - >/dev/null || true
pod hostpath-mount  "$OK_SC" "  volumes: [{name: h, hostPath: {path: /var/log}}]" "$OK_RES" | kubectl apply -f - >/dev/null || true
pod gpu-no-request  "$OK_SC" "" '{limits: {"amd.com/gpu": 1}, requests: {cpu: 10m}}' | kubectl apply -f - >/dev/null || true

note "waiting for the audit cycle ..."; sleep "${AUDIT_WAIT:-75}"

violations_for() {  # pod name -> count of audit violations that mention it
  if [[ "$ENGINE" == gatekeeper ]]; then
    kubectl get constraints -o json | python3 -c "import json,sys; d=json.load(sys.stdin); print(sum(1 for c in d['items'] for v in (c.get('status',{}).get('violations') or []) if v.get('name')=='$1'))"
  else
    kubectl get policyreport -n "$NS" -o json | python3 -c "import json,sys; d=json.load(sys.stdin); print(sum(1 for r in d['items'] for x in r.get('results',[]) if x.get('result')=='fail' and any(s.get('name')=='$1' for s in x.get('resources',[]) + r.get('scope') and [r['scope']] or [])))"
  fi
}
note ""; note "## Audit results"
for p in runs-as-root writable-rootfs hostpath-mount gpu-no-request; do
  n=$(violations_for "$p" || echo 0); check "$p" violation "$([[ "$n" -gt 0 ]] && echo violation || echo clean)"
done
n=$(violations_for compliant || echo 0); check compliant clean "$([[ "$n" -gt 0 ]] && echo violation || echo clean)"

note ""; note "## Enforcement (deny) of the hostPath rule"
if [[ "$ENGINE" == gatekeeper ]]; then
  C=$(kubectl get constraints -o name | grep -i hostpath | head -1)
  kubectl patch "$C" --type merge -p '{"spec":{"enforcementAction":"deny"}}' >/dev/null
  sleep 5
  if pod hostpath-denied "$OK_SC" "  volumes: [{name: h, hostPath: {path: /etc}}]" "$OK_RES" | kubectl apply -f - >/dev/null 2>&1; then check "deny hostPath" denied admitted; else check "deny hostPath" denied denied; fi
  kubectl patch "$C" --type merge -p '{"spec":{"enforcementAction":"dryrun"}}' >/dev/null
else
  kubectl patch clusterpolicy disallow-hostpath --type merge -p '{"spec":{"validationFailureAction":"Enforce"}}' >/dev/null
  sleep 5
  if pod hostpath-denied "$OK_SC" "  volumes: [{name: h, hostPath: {path: /etc}}]" "$OK_RES" | kubectl apply -f - >/dev/null 2>&1; then check "deny hostPath" denied admitted; else check "deny hostPath" denied denied; fi
  kubectl patch clusterpolicy disallow-hostpath --type merge -p '{"spec":{"validationFailureAction":"Audit"}}' >/dev/null
fi

note ""; note "## Summary"; note "$pass passed, $fail failed"
[[ $KEEP -eq 1 ]] || kubectl delete ns "$NS" --wait=false >/dev/null
[[ $fail -eq 0 ]]
