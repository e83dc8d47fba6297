#!/bin/bash
# HBM-side traffic per launch: two separate PMC passes (FETCH_SIZE, WRITE_SIZE), each with --kernel-trace only
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic_$c -- python $R/scratch/pmc_run.py > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/traffic_{c}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "tg::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[c] = (sum(v) / len(v), len(v))
print(json.dumps(out, indent=1))
PY
