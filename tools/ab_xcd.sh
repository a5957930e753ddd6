#!/bin/bash
# A/B of the XCD-aware placement of the object-SDF queries (8 meshes x 256 rows): kernel time (rocprofv3 stats) and
# fabric read traffic (rocprofv3 --pmc FETCH_SIZE, separate pass) of gq_sdf_wave_kernel, plain vs XCD-aware mapping.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/xs_$v /tmp/xp_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xs_$v -o run -- python3 $R/bench.py --no_cpu_baseline --n_objects 8 --event_steps 0 --sdf_plain_mapping $v > /tmp/xs_$v.json 2>/dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/xp_$v -o run -- python3 $R/bench.py --no_cpu_baseline --n_objects 8 --event_steps 0 --steps 40 --windows 1 --warmup 8 --sdf_plain_mapping $v > /dev/null 2>&1
  python3 - <<PY
import csv, json
rows=[r for r in csv.DictReader(open('/tmp/xs_$v/run_kernel_stats.csv')) if 'gq_sdf_wave' in r['Name']]
vals=[float(r['Counter_Value']) for r in csv.DictReader(open('/tmp/xp_$v/run_counter_collection.csv')) if r['Counter_Name']=='FETCH_SIZE' and 'gq_sdf_wave' in r['Kernel_Name']]
d=json.loads(open('/tmp/xs_$v.json').read().strip().splitlines()[-1])
print(json.dumps({"sdf_plain_mapping": $v, "evals_per_s": d["value"], "ms_per_step": d["ms_per_step"], "gq_sdf_wave_kernel_avg_us": float(rows[0]['AverageNs'])/1e3,
                  "fetch_bytes_per_launch": 2*1024*sum(vals)/max(len(vals),1), "launches_counted": len(vals)}))
PY
done
