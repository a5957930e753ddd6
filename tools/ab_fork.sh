#!/bin/bash
# A/B: fused two-role launches vs per-role launches on two graph branches, over the batch size (objects x 256 grasps)
out=gpurun_out/ab_fork.jsonl; : > $out
for no in 1 2 4 8 16; do
  for fl in "--fused 1" "--fused 0 --fork 1"; do
    python bench.py --no_cpu_baseline --n_objects $no --event_steps 0 $fl > gpurun_out/b_f.json 2>/dev/null
    python - "$no" "$fl" <<'PY' >> $out
import json, sys
d = json.loads(open("gpurun_out/b_f.json").read().strip().splitlines()[-1])
print(json.dumps({"n_objects": int(sys.argv[1]), "flags": sys.argv[2], "evals_per_s": round(d["value"]), "ms_per_step": round(d["ms_per_step"], 4)}))
PY
  done
done
cat $out
