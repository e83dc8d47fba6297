"""Per-wavefront means of rocprofv3 --pmc counter_collection.csv for the tg:: kernels (used by tools/profile_run.sh); without SQ_WAVES in the
pass: per-launch means."""
import collections
import csv
import glob
import sys

files = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "tg::" not in k:
        continue
    if v.get("SQ_WAVES"):
        w = sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"])
        print(k, {c: round(sum(x) / len(x) / w, 1) for c, x in v.items() if c != "SQ_WAVES"}, "waves", w, "launches", len(v["SQ_WAVES"]))
    else:
        print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "per launch; launches", len(next(iter(v.values()))))
