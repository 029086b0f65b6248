{{/* chart name, overridable */}}
{{- define "kserve-vllm-mini.name" -}}
{{- default .Chart.Name .Values.nameOverride | trunc 63 | trimSuffix "-" -}}
{{- end -}}

{{- define "kserve-vllm-mini.fullname" -}}
{{- if .Values.fullnameOverride -}}
{{- .Values.fullnameOverride | trunc 63 | trimSuffix "-" -}}
{{- else -}}
{{- $n := default .Chart.Name .Values.nameOverride -}}
{{- if contains $n .Release.Name -}}{{- .Release.Name | trunc 63 | trimSuffix "-" -}}{{- else -}}{{- printf "%s-%s" .Release.Name $n | trunc 63 | trimSuffix "-" -}}{{- end -}}
{{- end -}}
{{- end -}}

{{- define "kserve-vllm-mini.labels" -}}
helm.sh/chart: {{ printf "%s-%s" .Chart.Name .Chart.Version | replace "+" "_" }}
app.kubernetes.io/name: {{ include "kserve-vllm-mini.name" . }}
app.kubernetes.io/instance: {{ .Release.Name }}
app.kubernetes.io/version: {{ .Chart.AppVersion | quote }}
app.kubernetes.io/managed-by: {{ .Release.Service }}
{{- end -}}

{{/* namespace of the InferenceService: explicit > global > release */}}
{{- define "kserve-vllm-mini.isvcNamespace" -}}
{{- default (default .Release.Namespace .Values.global.isvc.namespace) .Values.inferenceService.namespace -}}
{{- end -}}

{{- define "kserve-vllm-mini.isvcName" -}}
{{- default .Values.inferenceService.name .Values.global.isvc.name -}}
{{- end -}}

{{- define "kserve-vllm-mini.s3SecretName" -}}
{{- default (printf "%s-s3" (include "kserve-vllm-mini.isvcName" .)) .Values.inferenceService.predictor.s3.secretName -}}
{{- end -}}

{{/* one ServingRuntime document; call with (dict "root" $ "rt" .Values.servingRuntimeX "format" "vllm" "gpu" true) */}}
{{- define "kserve-vllm-mini.runtime" -}}
apiVersion: serving.kserve.io/v1alpha1
kind: ServingRuntime
metadata:
  name: {{ .rt.name }}
  labels:
    {{- include "kserve-vllm-mini.labels" .root | nindent 4 }}
    {{- with .rt.labels }}{{ toYaml . | nindent 4 }}{{- end }}
  {{- with .rt.annotations }}
  annotations:
    {{- toYaml . | nindent 4 }}
  {{- end }}
spec:
  multiModel: false
  supportedModelFormats:
    - name: {{ .format }}
      autoSelect: {{ .autoSelect }}
  protocolVersions: [v2, openai]
  containers:
    - name: kserve-container
      image: {{ .rt.image | quote }}
      {{- with .rt.args }}
      args:
        {{- toYaml . | nindent 8 }}
      {{- end }}
      {{- with .rt.env }}
      env:
        {{- toYaml . | nindent 8 }}
      {{- end }}
      {{- if .gpu }}
      securityContext:
        allowPrivilegeEscalation: false
        capabilities:
          drop: [ALL]
        seccompProfile:
          type: RuntimeDefault
      {{- end }}
      resources:
        {{- toYaml .rt.resources | nindent 8 }}
{{- end -}}
