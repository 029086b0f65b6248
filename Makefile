SHELL := /bin/bash
PY ?= python3
CXX ?= g++
CHART := charts/kserve-vllm-mini
NS ?= ml-prod
RELEASE ?= demo
AIRGAP_DIR ?= airgap-bundle

.PHONY: help demo build build-native native test test-gpu lint fmt fmt-check typecheck yaml-lint validate-configs dashboards \
        helm-lint helm-install-minimal helm-install-with-kvmini helm-install-runtime helm-uninstall helm-package helm-test \
        airgap airgap-clean airgap-images airgap-charts airgap-wheels airgap-artifacts airgap-docs airgap-dashboards airgap-finalize dev-setup clean sanitize docs

help: ## Show this help
	@grep -E '^[a-zA-Z_-]+:.*## ' $(MAKEFILE_LIST) | awk 'BEGIN{FS=":.*## "}{printf "  %-26s %s\n", $$1, $$2}'

demo: ## Local end-to-end demo: mock server -> bench -> report (no cluster, no GPU)
	$(PY) -m kvmini bench --serve mock --service demo-llm --requests 100 --concurrency 10 --max-tokens 32 --report
	@echo "report: runs/latest/report.html"

native: ## Compile the HIP extension for gfx950 (in-tree .so)
	$(PY) build_native.py

build: ## Build the harness container image
	docker build -f Dockerfile.harness -t kvmini-harness:latest .

build-native: ## Build the serving image (native engine)
	docker build -f Dockerfile.native -t kvmini-native:latest .

SAN_DIR := build/sanitize
SAN_SRC := csrc/amdsmi_sampler.cpp csrc/tests/ring_stress.cpp
SAN_HOST := csrc/host_routines.cpp csrc/tests/host_routines_check.cpp
SAN_SECONDS ?= 3

sanitize: ## Host runtime (sampler ring, analysis routines) under ThreadSanitizer and ASan+UBSan: build + stress / self-check
	@mkdir -p $(SAN_DIR)
	$(CXX) -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=thread $(SAN_SRC) -lpthread -ldl -o $(SAN_DIR)/ring_stress_tsan
	$(CXX) -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=all $(SAN_SRC) -lpthread -ldl -o $(SAN_DIR)/ring_stress_asan
	$(CXX) -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=all $(SAN_HOST) -o $(SAN_DIR)/host_routines_asan
	$(CXX) -std=c++17 -O1 -g -fPIC -shared -fsanitize=address,undefined csrc/amdsmi_sampler.cpp csrc/host_routines.cpp -lpthread -ldl -o $(SAN_DIR)/libkvmini_rt_asan.so
	$(CXX) -std=c++17 -O1 -g -fPIC -shared -fsanitize=thread csrc/amdsmi_sampler.cpp csrc/host_routines.cpp -lpthread -ldl -o $(SAN_DIR)/libkvmini_rt_tsan.so
	TSAN_OPTIONS=halt_on_error=1 $(SAN_DIR)/ring_stress_tsan $(SAN_SECONDS)
	ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 $(SAN_DIR)/ring_stress_asan $(SAN_SECONDS)
	ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 $(SAN_DIR)/host_routines_asan
	@echo "sanitize: clean"

test: ## CPU test-suite
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu: ## GPU test-suite (needs an MI355X)
	$(PY) -u -m pytest tests -x -v -m gpu --timeout 120 --timeout-method thread

lint: ## ruff + shellcheck (when installed)
	@command -v ruff >/dev/null && ruff check kvmini tests bench.py || echo "ruff not installed; skipping"
	@command -v shellcheck >/dev/null && shellcheck *.sh tools/*.sh sweeps/*.sh runners/*.sh scripts/*.sh || echo "shellcheck not installed; skipping"

fmt: ## Auto-format
	@command -v ruff >/dev/null && ruff format kvmini tests bench.py && ruff check --fix kvmini tests bench.py || echo "ruff not installed"
	@command -v shfmt >/dev/null && shfmt -w -i 2 *.sh tools/*.sh || true

fmt-check: ## Formatting check only
	@command -v ruff >/dev/null && ruff format --check kvmini tests bench.py || echo "ruff not installed"

typecheck: ## mypy (when installed)
	@command -v mypy >/dev/null && mypy kvmini --ignore-missing-imports || echo "mypy not installed; skipping"

yaml-lint: ## Parse every YAML file outside Helm templates
	$(PY) -m kvmini validate --yaml-tree . --exclude charts/*/templates

validate-configs: ## Validate profiles, cost file, SLO file, reference matrix, chart values vs schema
	$(PY) -m kvmini validate --all

docs: ## Static documentation site from README + docs/ (site/index.html)
	$(PY) -m kvmini docs-site --out site

dashboards: ## Regenerate dashboards/*.json
	$(PY) -m kvmini.report.dashboards --out dashboards

helm-lint: ## Lint both charts
	helm dependency update $(CHART)
	helm lint charts/kvmini
	helm lint $(CHART) -f $(CHART)/examples/values-minimal.yaml

helm-install-minimal: ## InferenceService only
	helm upgrade --install $(RELEASE) $(CHART) -n $(NS) --create-namespace -f $(CHART)/examples/values-minimal.yaml

helm-install-with-kvmini: ## InferenceService + harness pod
	helm dependency update $(CHART)
	helm upgrade --install $(RELEASE) $(CHART) -n $(NS) --create-namespace -f $(CHART)/examples/values-with-kvmini.yaml

helm-install-runtime: ## Also install the vLLM-ROCm and native ServingRuntimes
	helm upgrade --install $(RELEASE) $(CHART) -n $(NS) --create-namespace -f $(CHART)/examples/values-runtime.yaml

helm-uninstall: ## Remove the release
	helm uninstall $(RELEASE) -n $(NS)

helm-package: ## Package the parent chart into dist/
	helm dependency update $(CHART)
	mkdir -p dist && helm package $(CHART) -d dist

helm-test: ## helm test (smoke pod curls the endpoint)
	helm test $(RELEASE) -n $(NS)

airgap: airgap-images airgap-artifacts airgap-examples airgap-docs airgap-dashboards airgap-finalize ## Complete air-gapped bundle

airgap-clean: ## Remove the bundle directory
	rm -rf $(AIRGAP_DIR)

airgap-images: airgap-clean ## Save container images (harness, native engine, vLLM-ROCm)
	mkdir -p $(AIRGAP_DIR)/images
	@for img in kvmini-harness:latest kvmini-native:latest rocm/vllm:latest; do \
	  if docker image inspect $$img >/dev/null 2>&1; then docker save $$img | gzip > $(AIRGAP_DIR)/images/$$(echo $$img | tr '/:' '__').tar.gz; echo "saved $$img"; \
	  else echo "image $$img not present locally; skipping"; fi; done

airgap-charts: ## Package charts into the bundle
	mkdir -p $(AIRGAP_DIR)/charts
	@command -v helm >/dev/null && helm package charts/kvmini -d $(AIRGAP_DIR)/charts && helm dependency update $(CHART) && helm package $(CHART) -d $(AIRGAP_DIR)/charts || cp -r charts $(AIRGAP_DIR)/charts/src

airgap-wheels: ## Download Python wheels
	mkdir -p $(AIRGAP_DIR)/wheels
	$(PY) -m pip wheel . -w $(AIRGAP_DIR)/wheels || echo "wheel build needs an index or a wheelhouse; skipping"

airgap-artifacts: airgap-charts airgap-wheels ## Charts + wheels + configs
	mkdir -p $(AIRGAP_DIR)/config
	cp -r cost.yaml slo.json reference-matrix.yaml isvc.yaml example-config.yaml runners profiles policies $(AIRGAP_DIR)/config/

airgap-examples: ## Example configuration and a run script that works from the bundle alone
	mkdir -p $(AIRGAP_DIR)/examples
	cp example-config.yaml cost.yaml slo.json $(AIRGAP_DIR)/examples/
	cp runners/profiles/standard.yaml runners/profiles/tool-calling.yaml $(AIRGAP_DIR)/examples/ 2>/dev/null || true
	$(PY) -m kvmini.deploy.cli --backend native --model-name llama-3-8b --service llama3-8b-demo --namespace demo --dry-run | sed -n '/^apiVersion:/,$$p' > $(AIRGAP_DIR)/examples/isvc-native.yaml
	printf '%s\n' '#!/usr/bin/env bash' '# Deploy the native engine from the bundled images and benchmark it; no registry, no index.' 'set -euo pipefail' \
	  'here="$$(cd "$$(dirname "$$0")" && pwd)"' \
	  'for f in "$$here"/../images/*.tar.gz; do [ -e "$$f" ] && gunzip -c "$$f" | docker load; done' \
	  'kubectl apply -f "$$here/isvc-native.yaml"' \
	  'kubectl wait --for=condition=Ready inferenceservice/llama3-8b-demo -n demo --timeout=900s' \
	  'kvmini bench --namespace demo --service llama3-8b-demo --model llama-3-8b --requests 100 --concurrency 10 --cost-file "$$here/cost.yaml"' \
	  'kvmini gate --slo "$$here/slo.json" --results "$$(ls -dt runs/*/ | head -1)results.json"' \
	  'echo "results under runs/"' > $(AIRGAP_DIR)/examples/run-benchmark.sh
	chmod +x $(AIRGAP_DIR)/examples/run-benchmark.sh

airgap-docs: ## Documentation
	mkdir -p $(AIRGAP_DIR)/docs && cp -r README.md docs/* $(AIRGAP_DIR)/docs/ 2>/dev/null || true

airgap-dashboards: ## Grafana dashboards
	mkdir -p $(AIRGAP_DIR)/dashboards && cp dashboards/*.json $(AIRGAP_DIR)/dashboards/

airgap-finalize: ## Checksums + tarball
	cd $(AIRGAP_DIR) && find . -type f ! -name SHA256SUMS -print0 | sort -z | xargs -0 sha256sum > SHA256SUMS
	tar -czf $(AIRGAP_DIR).tar.gz $(AIRGAP_DIR) && echo "bundle: $(AIRGAP_DIR).tar.gz"

install-deps: ## Python dependencies of the harness (the engine's torch comes with the ROCm image)
	$(PY) -m pip install -e .
	@if [ -f requirements-dev.txt ]; then $(PY) -m pip install -r requirements-dev.txt; else $(PY) -m pip install -e ".[dev]" || echo "(no dev extras installed)"; fi

install-tools: ## Developer tools: ruff, mypy, yamllint (pip); shellcheck, shfmt, helm (system package manager)
	$(PY) -m pip install ruff mypy yamllint pytest pytest-timeout || true
	@if command -v apt-get >/dev/null 2>&1; then sudo apt-get update && sudo apt-get install -y shellcheck shfmt; \
	elif command -v dnf >/dev/null 2>&1; then sudo dnf install -y ShellCheck shfmt; \
	elif command -v brew >/dev/null 2>&1; then brew install shellcheck shfmt helm; \
	else echo "install shellcheck, shfmt and helm with your package manager"; fi
	@command -v hipcc >/dev/null 2>&1 || echo "hipcc not on PATH: 'make native' needs ROCm (/opt/rocm/bin)"

dev-setup: ## Editable install + native build
	$(PY) -m pip install -e ".[dev]" || true
	$(PY) build_native.py

clean: ## Remove build artifacts
	rm -rf build dist *.egg-info .pytest_cache $(AIRGAP_DIR) $(AIRGAP_DIR).tar.gz
	find . -name __pycache__ -type d -prune -exec rm -rf {} +
