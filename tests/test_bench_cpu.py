"""bench.py's bookkeeping that needs no GPU: what `roofline.traffic` may quote, and the algorithmic bytes behind `roofline.achieved`."""
import json
import os

import bench


def test_traffic_is_quoted_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    """A PMC file measured on other sources (or on another configuration) is never quoted as this binary's traffic (a later
    binary silently quoted an older measurement)."""
    have = bench.source_hash()
    wl = {"env": "edge_follow-v0", "num_envs": 1024, "image_size": 128, "physics": "f64",
          "k_render_tactile": {"kernel": "k_render_blocks<16>", "fetch_corrected_kb": 1000.0, "write_kb": 3000.0}}
    f = tmp_path / "traffic.json"
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
    ab = bench.algo_bytes("edge_follow-v0", 128)
    f.write_text(json.dumps({"source_sha16": "0" * 16, "workloads": [wl]}))
    stale = bench.read_traffic("edge_follow-v0", 1024, 128, "k_render_tactile", ab)
    assert stale["bytes_per_launch"] is None and have in stale["stale"]
    f.write_text(json.dumps({"source_sha16": have, "workloads": [wl]}))
    ok = bench.read_traffic("edge_follow-v0", 1024, 128, "k_render_tactile", ab)
    assert ok["bytes_per_launch"] == 4000 * 1024 and ok["source_sha16"] == have
    assert abs(ok["vs_algorithmic"] - 4000 * 1024 / (ab * 1024)) < 1e-3
    assert bench.read_traffic("edge_follow-v0", 2048, 128, "k_render_tactile", ab) is None      # another batch size: nothing to quote
    assert bench.read_traffic("object_push-v0", 1024, 128, "k_render_tactile", ab) is None
    f.write_text("not json")
    assert bench.read_traffic("edge_follow-v0", 1024, 128, "k_render_tactile", ab) is None


def test_source_hash_covers_every_file_the_library_is_built_from():
    """The hash moves when a kernel source or the C-ABI header moves, and only then."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(bench.ROOT, "tactile_gym_amd", "csrc", "*")) + glob.glob(os.path.join(bench.ROOT, "include", "*.h")))
    names = {os.path.basename(p) for p in files}
    assert {"tg_raster.hip", "tg_api.hip", "tg_contact_wave.hip", "tg_exchange.hip", "tg_scene.hip", "tg_noise.hip", "tg_kernels.hpp",
            "tg_physics.hpp", "build.sh", "tactile_gym_hip.h"} <= names
    h = hashlib.sha256()
    for p in files:
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode() + b"\0" + open(p, "rb").read())
    assert bench.source_hash() == h.hexdigest()[:16]


def test_algorithmic_bytes_follow_survey_8d():
    """one uint8 image written per env step plus the state / action / reward words (+ the heightfield read for surface_follow,
    + the extended feature for object_push)."""
    assert bench.algo_bytes("edge_follow-v0", 128) == 128 * 128 + 216
    assert bench.algo_bytes("object_balance-v0", 256) == 256 * 256 + 300
    assert bench.algo_bytes("surface_follow-v0", 128) > 2 * 128 * 128
    assert bench.algo_bytes("object_push-v0", 128) == 128 * 128 + 400
