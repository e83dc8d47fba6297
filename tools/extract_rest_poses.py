#!/usr/bin/env python3
"""Golden vectors from the reference's rest-pose tables (run in the build container, where /root/reference exists).

Every entry of the five `rest_poses.py` tables (rl_envs/exploration/{edge_follow,surface_follow}/rest_poses.py,
rl_envs/nonprehensile_manipulation/{object_push,object_balance,object_roll}/rest_poses.py) is a PyBullet-produced joint vector
(indexed by URDF joint, fixed joints included) for a known TCP pose: the pose the env's reset drives to.  They are data; the
modules import nothing but numpy, so they are imported here and their UR5 / MG400 rows written to tests/golden/rest_poses.json
(Franka / Kuka rows are out of scope).  tests/test_oracle_golden.py runs them through urdf_compile + FK.
"""
import importlib.util
import json
import os

REF = os.environ.get("TG_REFERENCE", "/root/reference/tactile_gym")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLES = {
    "edge_follow": "rl_envs/exploration/edge_follow/rest_poses.py",
    "surface_follow": "rl_envs/exploration/surface_follow/rest_poses.py",
    "object_push": "rl_envs/nonprehensile_manipulation/object_push/rest_poses.py",
    "object_balance": "rl_envs/nonprehensile_manipulation/object_balance/rest_poses.py",
    "object_roll": "rl_envs/nonprehensile_manipulation/object_roll/rest_poses.py",
}
SENSORS = ("tactip", "digit", "digitac")


def main():
    out = {}
    for table, rel in TABLES.items():
        spec = importlib.util.spec_from_file_location(f"_rest_{table}", os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        rows = []
        for arm in ("ur5", "mg400"):
            d = mod.rest_poses_dict.get(arm, {})
            for k, v in d.items():
                if k in SENSORS:                       # [arm][sensor][type]
                    for typ, vec in v.items():
                        rows.append(dict(arm=arm, sensor=k, type=typ, joints=[float(x) for x in vec]))
                elif isinstance(v, dict):
                    continue                           # a Franka / Kuka table nested under the arm by an upstream indentation slip
                else:                                  # [arm][type]: the table does not distinguish sensors (balance, roll)
                    rows.append(dict(arm=arm, sensor=None, type=k, joints=[float(x) for x in v]))
        out[table] = dict(source=rel, rows=rows)
    path = os.path.join(ROOT, "tests", "golden", "rest_poses.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.relpath(path, ROOT), {k: len(v["rows"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
