#!/bin/bash
# A/B: LDS list capacities of the stand-alone penetration query (512 / 256 / 128 entries per block) over the batch size
out=gpurun_out/ab_caps.jsonl; : > $out
for no in 2 8 16; do
  for caps in 1 2 3; do
    python bench.py --no_cpu_baseline --n_objects $no --event_steps 0 --pen_caps $caps > gpurun_out/b_f.json 2>/dev/null
    python - "$no" "$caps" <<'PY' >> $out
import json, sys
d = json.loads(open("gpurun_out/b_f.json").read().strip().splitlines()[-1])
print(json.dumps({"n_objects": int(sys.argv[1]), "pen_caps": int(sys.argv[2]), "evals_per_s": round(d["value"]), "ms_per_step": round(d["ms_per_step"], 4),
                  "query_ms": round(d["roofline"]["kernel_ms"], 4), "ranked_inline": d["executed_work"]["hand_pen"]["pairs_ranked_inline"],
                  "entries": d["executed_work"]["hand_pen"]["pairs_reaching_candidates"]}))
PY
  done
done
cat $out
