#!/bin/bash
# A/B of the in-kernel weight prefetch (KVMINI_PREFETCH, kvmini/ops/hip_ops.py::prefetch_plan): the
# decode step of the headline shape (Llama-3-8B, batch 20, HIP graphs) under several settings,
# one process each (the setting is read once per process).  Usage, on the GPU box:
#   scripts/gpu_prefetch_ab.sh [out_dir] [spec ...]      (a spec of "0" = prefetch off)
set -u
out=${1:-gpurun_out/prefetch_ab}
shift || true
mkdir -p "$out"
specs=("$@")
if [ ${#specs[@]} -eq 0 ]; then
  specs=("0" "qkv=16:1536,o=16:64,gu=12:1792,head=4:1024" "qkv=16:1536" "o=16:64" "gu=12:1792")
fi
i=0
for spec in "${specs[@]}"; do
  echo "== KVMINI_PREFETCH=$spec"
  KVMINI_PREFETCH="$spec" timeout -k 10 240 python scripts/gpu_microbench.py --what step --kernels hip --out "$out/step_$i.json" > "$out/step_$i.log" 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then
    echo "step bench failed (rc $rc) for spec $spec"; tail -5 "$out/step_$i.log"; exit $rc
  fi
  python - "$out/step_$i.json" "$spec" <<'EOF'
import json, sys
rows = json.load(open(sys.argv[1]))
for r in rows:
    print(json.dumps({"prefetch": sys.argv[2], "ms_per_step": round(r["ms_per_step"], 4), "eff_TBps": round(r["eff_TBps"], 3)}))
EOF
  i=$((i + 1))
done | tee "$out/summary.txt"
