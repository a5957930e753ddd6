#!/bin/bash
# A/B: LDS list capacities of the stand-alone penetration query (512 / 256 entries per block) x launch mode over the batch size
out=gpurun_out/ab_caps.jsonl; : > $out
for no in 2 8 16; do
  for caps in 1 2; do
    for fl in "--fork 1 --fused 0" "--fork 0 --fused 0"; do
    python bench.py --no_cpu_baseline --n_objects $no --event_steps 0 --pen_caps $caps $fl > gpurun_out/b_f.json 2>/dev/null
    python - "$no" "$caps" "$fl" <<'PY' >> $out
import json, sys
d = json.loads(open("gpurun_out/b_f.json").read().strip().splitlines()[-1])
print(json.dumps({"n_objects": int(sys.argv[1]), "pen_caps": int(sys.argv[2]), "mode": d["config"]["branches"], "evals_per_s": round(d["value"]), "ms_per_step": round(d["ms_per_step"], 4),
                  "query_ms": round(d["roofline"]["kernel_ms"], 4), "ranked_inline": d["executed_work"]["hand_pen"]["pairs_ranked_inline"]}))
PY
    done
  done
done
cat $out
