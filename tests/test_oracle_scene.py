"""CPU checks of the scene-camera oracle (oracle/minibullet.c mb_render_scene, minibullet.py scene_view_matrix): SURVEY 8 row f4.

Upstream's pixels cannot be pinned (its renderer is the GL driver's and the checkout holds no scene image), so what is pinned here is
the camera geometry - against closed-form pinhole projections written independently in this file - and the properties the device
raster relies on (order independence, no cracks along shared edges, near-plane crossing triangles)."""
import math

import numpy as np

from oracle import minibullet as mb

CAM = ([0.35, 0.0, -0.25], 0.75, 90.0, -35.0, 75.0, 0.1, 100.0)        # edge_follow_env.py:187-195


def _eye(cam):
    """Independent statement of the eye point: `distance` behind the target along the viewing direction given by yaw (about +z, zero
    looking along +y) and pitch (negative = looking down)."""
    t, d, yaw, pitch = np.asarray(cam[0]), cam[1], math.radians(cam[2]), math.radians(cam[3])
    fwd = np.array([-math.sin(yaw) * math.cos(pitch), math.cos(yaw) * math.cos(pitch), math.sin(pitch)])
    return t - d * fwd, fwd


def _project(cam, p, W, H):
    """Pinhole projection of world point p to (column, row) pixel coordinates, from first principles."""
    eye, fwd = _eye(cam)
    right = np.cross(fwd, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    d = np.asarray(p) - eye
    f = (H / 2.0) / math.tan(math.radians(cam[4]) / 2.0)
    return W / 2.0 + f * (d @ right) / (d @ fwd), H / 2.0 - f * (d @ up) / (d @ fwd)


def test_view_matrix_known_answers():
    V, t = mb.scene_view_matrix(*CAM[:4])
    eye, fwd = _eye(CAM)
    assert np.allclose(eye, [0.35 + 0.75 * math.cos(math.radians(35)), 0.0, -0.25 + 0.75 * math.sin(math.radians(35))], atol=1e-12)
    assert np.allclose(V @ eye + t, 0.0, atol=1e-12)                                   # the eye is the origin of eye space
    assert np.allclose(V @ np.asarray(CAM[0]) + t, [0.0, 0.0, -0.75], atol=1e-12)      # the target sits on the -z axis at `distance`
    assert np.allclose(V @ V.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(V) - 1.0) < 1e-12
    assert (V @ np.array([0.0, 0.0, 1.0]))[1] > 0.0                                     # world up points up in the image


def _marker(p, r=0.004):
    """A small octahedron around p."""
    p = np.asarray(p, dtype=np.float64)
    v = np.array([p + [r, 0, 0], p - [r, 0, 0], p + [0, r, 0], p - [0, r, 0], p + [0, 0, r], p - [0, 0, r]])
    t = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]])
    return v, t


def _render(verts, tris, rgb, cam, n, frames=None, frame_ids=None):
    frames = frames or [(np.eye(3), np.zeros(3))]
    frame_ids = np.zeros(len(tris), np.uint8) if frame_ids is None else frame_ids
    return mb.render_scene(verts, tris, frame_ids, np.tile(np.asarray(rgb, np.uint8), (len(tris), 1)), frames,
                           mb.scene_view_matrix(*cam[:4]), (-50.0, 30.0, 100.0), cam[4], cam[5], cam[6], n, n, (1, 2, 3))


def test_markers_land_on_the_closed_form_pixels():
    rng = np.random.default_rng(0)
    for cam in (CAM, ([0.1, 0.0, -0.35], 1.0, 90.0, -45.0, 75.0, 0.1, 100.0), ([-0.1, 0.0, 0.25], 1.0, 90.0, -10.0, 75.0, 0.1, 100.0),
                ([0.16, 0.0, 0.14], 0.45, -2.0, -30.0, 75.0, 0.1, 100.0)):
        for _ in range(6):
            p = np.asarray(cam[0]) + rng.uniform(-0.12, 0.12, 3)
            v, t = _marker(p, 0.01)
            img = _render(v, t, (255, 0, 0), cam, 256)
            hit = np.argwhere(img[..., 0] > 100)
            assert len(hit) > 0
            col, row = _project(cam, p, 256, 256)
            assert abs(hit[:, 1].mean() + 0.5 - col) < 1.0 and abs(hit[:, 0].mean() + 0.5 - row) < 1.0
            assert (img[img[..., 0] <= 100] == (1, 2, 3)).all()                         # everything else is background


def test_plane_crossing_the_near_plane_and_its_horizon():
    """The reference's 200 m ground plane passes under and behind the camera: no clipping artefacts, and it ends at the horizon row."""
    v = np.array([[100.0, -100.0, -0.625], [100.0, 100.0, -0.625], [-100.0, 100.0, -0.625], [-100.0, -100.0, -0.625]])
    t = np.array([[0, 1, 2], [0, 2, 3]])
    n = 128
    img = _render(v, t, (255, 255, 255), CAM, n)
    ground = img[..., 0] > 100
    eye, _ = _eye(CAM)
    _, row_far = _project(CAM, [-100.0, 0.0, -0.625], n, n)                            # the far edge straight ahead
    rows = np.arange(n) + 0.5
    assert (ground[:, n // 2] == (rows > row_far)).all()
    assert ground[-1].all() and not ground[0].any()
    assert (np.diff(ground.astype(int), axis=0) >= 0).all()                             # no holes: once ground starts it continues downwards


def test_triangle_order_does_not_matter_and_shared_edges_have_no_cracks():
    rng = np.random.default_rng(3)
    # a tessellated, tilted sheet filling the view + random small triangles in front of it
    g = 24
    xs, ys = np.meshgrid(np.linspace(-0.6, 0.9, g), np.linspace(-0.7, 0.7, g), indexing="ij")
    v = np.stack([xs.ravel(), ys.ravel(), -0.4 + 0.1 * xs.ravel() + 0.05 * np.sin(7 * ys.ravel())], 1)
    t = []
    for i in range(g - 1):
        for j in range(g - 1):
            a = i * g + j
            t += [(a, a + g, a + g + 1), (a, a + g + 1, a + 1)]
    t = np.array(t)
    extra_v = np.asarray(CAM[0]) + rng.uniform(-0.3, 0.3, (300, 3))
    extra_t = np.arange(300).reshape(100, 3) + len(v)
    verts, tris = np.concatenate([v, extra_v]), np.concatenate([t, extra_t])
    rgb = rng.integers(0, 256, (len(tris), 3)).astype(np.uint8)
    view = mb.scene_view_matrix(*CAM[:4])

    def draw(order):
        return mb.render_scene(verts, tris[order], np.zeros(len(tris), np.uint8), rgb[order], [(np.eye(3), np.zeros(3))], view,
                               (-50.0, 30.0, 100.0), 75.0, 0.1, 100.0, 128, 128, (1, 2, 3))
    ref = draw(np.arange(len(tris)))
    for k in range(3):
        assert np.array_equal(ref, draw(rng.permutation(len(tris))))
    # the sheet alone covers every pixel it spans: no background shows through along shared edges
    sheet = mb.render_scene(v, t, np.zeros(len(t), np.uint8), np.full((len(t), 3), 200, np.uint8), [(np.eye(3), np.zeros(3))], view,
                            (-50.0, 30.0, 100.0), 75.0, 0.1, 100.0, 128, 128, (1, 2, 3))
    covered = sheet[..., 0] != 1
    assert covered.mean() > 0.5
    hole = ~covered[1:-1, 1:-1] & covered[:-2, 1:-1] & covered[2:, 1:-1] & covered[1:-1, :-2] & covered[1:-1, 2:]
    assert not hole.any()                                 # a background pixel enclosed by covered ones would be a crack


def test_frames_move_their_triangles_and_shading_is_two_sided():
    v, t = _marker([0.0, 0.0, 0.0], 0.03)
    p1, p2 = np.asarray(CAM[0]) + [0.0, 0.1, 0.05], np.asarray(CAM[0]) + [0.0, -0.1, 0.05]
    verts, tris = np.concatenate([v, v]), np.concatenate([t, t[:, ::-1] + len(v)])        # second copy wound the other way
    ids = np.concatenate([np.ones(len(t), np.uint8), np.full(len(t), 2, np.uint8)])
    img = _render(verts, tris, (200, 200, 200), CAM, 128, frames=[(np.eye(3), np.zeros(3)), (np.eye(3), p1), (np.eye(3), p2)], frame_ids=ids)
    lit = np.argwhere(img[..., 0] > 100)
    left, right = lit[lit[:, 1] < 64], lit[lit[:, 1] >= 64]
    assert len(left) >= 12 and abs(len(left) - len(right)) <= 2                                       # mirror images about the centre column
    c1, r1 = _project(CAM, p1, 128, 128)
    assert abs(left[:, 1].mean() + 0.5 - c1) < 1.0 or abs(right[:, 1].mean() + 0.5 - c1) < 1.0
    vals = set(np.unique(img[lit[:, 0], lit[:, 1], 0]))
    assert all(120 <= x <= 190 for x in vals)             # 200 x (0.6 .. 0.95): ambient + diffuse, whichever way the triangle is wound


def test_oracle_env_visual_observation(edge_modes):
    from oracle.ref_env import OracleEdgeFollowEnv
    modes = dict(edge_modes, observation_mode="visuotactile")
    e = OracleEdgeFollowEnv(seed=1, image_size=(64, 64), env_modes=modes)
    obs = e.reset()
    assert set(obs) == {"tactile", "visual"} and obs["visual"].shape == (64, 64, 3) and obs["visual"].dtype == np.uint8
    img = obs["visual"]
    blue = (img[..., 2] > 150) & (img[..., 0] < 30)                                         # the edge stimulus (rgba 0 0 1 1)
    col, row = _project(CAM, e.edge_pos, 64, 64)
    rr, cc = np.where(blue)
    assert blue.sum() > 20 and abs(cc.mean() - col) < 6 and abs(rr.mean() - row) < 6
    wood = (img[..., 0] > 200) & (img[..., 2] < 200) & (img[..., 2] > 100)                  # the table top (1.0 0.9 0.75)
    assert wood.mean() > 0.4
    o2, _, _, _ = e.step(np.array([0.25, 0.25], np.float32))
    assert (o2["visual"] != img).any()


def _blend(spheres, cam=CAM, n=128, tri=None):
    """A scene that is empty (or one big opaque triangle) plus translucent spheres."""
    if tri is None:
        verts, tris = np.zeros((3, 3)), np.zeros((0, 3), np.int32)
    else:
        verts, tris = np.asarray(tri, dtype=np.float64), np.array([[0, 1, 2]], np.int32)
    return mb.render_scene(verts, tris, np.zeros(len(tris), np.uint8), np.tile(np.array([[0, 0, 255]], np.uint8), (len(tris), 1)), [(np.eye(3), np.zeros(3))],
                           mb.scene_view_matrix(*cam[:4]), (-50.0, 30.0, 100.0), cam[4], cam[5], cam[6], n, n, (10, 20, 30), spheres=spheres)


def test_translucent_spheres_known_answers():
    """mb_blend_spheres (goal indicators / trajectory markers, PARITY A33b) against first principles: where the disc lands and how big it
    is (pinhole projection), the blend rule out = u8(alpha src + (1 - alpha) dst + 0.5) with src inside the shading range, occlusion by an
    opaque surface in front, visibility in front of one behind, and list-order blending of two overlapping spheres."""
    n, c, r = 128, np.asarray(CAM[0]) + [0.0, 0.02, 0.03], 0.02
    img = _blend([(c, r, (255.0, 0.0, 0.0), 0.5)])
    hit = (img != np.array([10, 20, 30])).any(axis=2)
    col, row = _project(CAM, c, n, n)
    ys, xs = np.nonzero(hit)
    assert abs(xs.mean() + 0.5 - col) < 0.6 and abs(ys.mean() + 0.5 - row) < 0.6                # the disc's centre of mass
    eye, fwd = _eye(CAM)
    f = (n / 2.0) / math.tan(math.radians(CAM[4]) / 2.0)
    r_px = f * r / ((c - eye) @ fwd)
    assert abs(hit.sum() - math.pi * r_px ** 2) < 0.12 * math.pi * r_px ** 2 + 8             # its area (perspective stretches it a little)
    px = img[int(row), int(col)].astype(float)
    # red channel: half of a shaded 255 (0.6 .. 0.95 of it) + half of the background; green / blue: half of the background, rounded
    assert 0.5 * 0.6 * 255 + 5 - 1 <= px[0] <= 0.5 * 0.95 * 255 + 5 + 1 and px[1] == 10 and px[2] == 15
    # an opaque triangle in front of the sphere hides it; behind the sphere it shows through at half weight
    eye_to_c = (c - eye) / np.linalg.norm(c - eye)
    right = np.cross(eye_to_c, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right)
    up = np.cross(right, eye_to_c)

    def wall(dist):
        o = eye + dist * eye_to_c
        return [o - 2 * right - 2 * up, o + 2 * right - 2 * up, o + 3 * up]
    front, back = _blend([(c, r, (255.0, 0.0, 0.0), 0.5)], tri=wall(0.3)), _blend([(c, r, (255.0, 0.0, 0.0), 0.5)], tri=wall(1.2))
    assert np.array_equal(front, _blend([], tri=wall(0.3)))                                       # nothing of the sphere in front of the wall's pixels
    wall_only = _blend([], tri=wall(1.2))
    d = (back != wall_only).any(axis=2)
    assert d.sum() == hit.sum() and np.array_equal(d, hit)                                        # the same disc, now over the wall
    b = back[int(row), int(col)].astype(float); w0 = wall_only[int(row), int(col)].astype(float)
    assert b[1] == 0 and abs(b[2] - math.floor(0.5 * w0[2] + 0.5)) <= 0 and b[0] == px[0] - 5     # blue wall at half weight under the same red
    # two spheres, one behind the other along the ray: both blend (no depth between translucent fragments), the later one on top
    c2 = c + 0.05 * eye_to_c
    ab = _blend([(c, r, (255.0, 0.0, 0.0), 0.5), (c2, r, (0.0, 255.0, 0.0), 0.5)])
    ba = _blend([(c2, r, (0.0, 255.0, 0.0), 0.5), (c, r, (255.0, 0.0, 0.0), 0.5)])
    pa, pb = ab[int(row), int(col)].astype(int), ba[int(row), int(col)].astype(int)
    assert pa[1] > pa[0] and pb[0] > pb[1] and not np.array_equal(ab, ba)
    # alpha 0 slots are ignored
    assert np.array_equal(_blend([(c, r, (255.0, 0.0, 0.0), 0.0)]), _blend([]))


def test_oracle_envs_draw_their_goal_markers():
    """Every task env lists its translucent visuals from the reference's own code paths (scene_spheres): edge / surface / roll one goal
    sphere, object_push ten trajectory markers with the current target blue, object_balance none - and they change the picture."""
    import warnings
    from oracle import ref_env
    warnings.simplefilter("ignore")
    base = dict(control_mode="TCP_velocity_control", observation_mode="visuotactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
    cases = [("OracleEdgeFollowEnv", dict(base, movement_mode="xy", noise_mode="rand_height"), 1),
             ("OracleSurfaceFollowAutoEnv", dict(base, movement_mode="xyzRxRy", noise_mode="simplex"), 1),
             ("OracleObjectPushEnv", dict(base, movement_mode="TyRz", rand_init_orn=False, rand_obj_mass=False, traj_type="simplex", observation_mode="tactile_and_feature"), 10),
             ("OracleObjectRollEnv", dict(base, movement_mode="xy", rand_init_obj_pos=True, rand_obj_size=True, rand_embed_dist=True, observation_mode="tactile_and_feature"), 1),
             ("OracleObjectBalanceEnv", dict(base, movement_mode="xy", object_mode="pole", rand_gravity=True, rand_embed_dist=True, observation_mode="tactile"), 0)]
    for cls, modes, count in cases:
        o = getattr(ref_env, cls)(seed=3, image_size=(128, 128), env_modes=modes)
        o.reset()
        sp = o.scene_spheres()
        assert len(sp) == count + 1 and sp[0][1] == 0.001 and sp[0][2] == (229.5, 0.0, 51.0), cls      # slot 0: the arm's TCP marker
        sp = sp[1:]
        if cls == "OracleObjectPushEnv":
            g = o.targ_traj_list_id            # the current target blue, the ones already passed red, the rest green (object_push_env.py:281-282, 360-366)
            assert [s[2] for s in sp] == [(255.0, 0.0, 0.0)] * g + [(0.0, 0.0, 255.0)] + [(0.0, 255.0, 0.0)] * (9 - g)
        if o.scene_camera() is None:          # the object envs' scene cameras live with the product (rl_envs/*.py scene_spec); tests/test_gpu_visual.py draws them
            continue
        with_markers = o.visual_image()
        o.scene_spheres = lambda: []
        assert (with_markers != o.visual_image()).any(), cls
