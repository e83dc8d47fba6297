"""Copies the evidence of tools/profile_run.sh from gpurun_out/<tag>/ into profiles/<tag>_* and assembles profiles/<round>_traffic.json (what
bench.py quotes as roofline.traffic) with the hash of the sources the measured library was built from.  usage: profile_collect.py [tag]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r6_final"
src = os.path.join(ROOT, "gpurun_out", tag)
for f in sorted(glob.glob(os.path.join(src, "*"))):
    if os.path.isfile(f) and not f.endswith(".err"):
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_{os.path.basename(f)}"))
workloads = []
for name, env, size in (("edge", "edge_follow-v0", 128), ("object_push-v0", "object_push-v0", 128), ("object_balance-v0", "object_balance-v0", 256),
                        ("surface_follow-v0", "surface_follow-v0", 128)):
    p = os.path.join(src, f"traffic_{name}.json")
    if not os.path.isfile(p):
        continue
    t = json.load(open(p))
    wl = {"env": env, "num_envs": 1024, "image_size": size, "physics": "f64", "algorithmic_kb_per_launch": round(bench.algo_bytes(env, size) * 1024 / 1024.0, 1)}     # bytes per env step x 1024 envs, in KB
    for k, v in t.items():
        short = k.split("<")[0].replace("void ", "").replace("tg::", "").replace("(anonymous namespace)::", "").strip()
        if short.startswith("k_step"):
            wl["k_step"] = dict(v, kernel=k.replace("void ", "").replace("tg::", ""))
        elif short.startswith("k_render") and v["write_kb"] * v["launches"] >= wl.get("k_render_tactile", {}).get("_total", 0):
            wl["k_render_tactile"] = dict(v, kernel=k.replace("void ", "").replace("tg::", ""), _total=v["write_kb"] * v["launches"])
    if "k_step" in wl and "k_render_tactile" in wl:
        # the render kernel is launched twice per step where finished envs are re-drawn after their reset (a full and a masked launch): the
        # figures below are per STEP (mean per launch x launches per step), comparable with the algorithmic bytes of one step
        r, per_step = wl["k_render_tactile"], wl["k_render_tactile"]["launches"] / max(wl["k_step"]["launches"], 1)
        r.pop("_total", None)
        if per_step > 1.2:
            for key in ("fetch_kb", "fetch_corrected_kb", "write_kb"):
                r[key] = round(r[key] * per_step, 1)
            r["launches_per_step"] = round(per_step, 2)
    workloads.append(wl)
out = {"_what": "HBM-side traffic per kernel launch from rocprofv3 PMC (separate passes: --pmc FETCH_SIZE, --pmc WRITE_SIZE, each with --kernel-trace; "
                "tools/profile_run.sh traffic(), parsed by tools/traffic_parse.py), 1024 envs, f64, default solver; values in KB as reported.  Per "
                "MI355X_MICROARCH.md (HBM section) FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950: fetch_corrected_kb doubles it.",
       "source_sha16": open(os.path.join(src, "source_sha16.txt")).read().strip(), "tag": tag, "workloads": workloads}
name = tag.split("_")[0] + "_traffic.json"
json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print("wrote profiles/" + name + " for sources", out["source_sha16"], [w["env"] for w in workloads])
