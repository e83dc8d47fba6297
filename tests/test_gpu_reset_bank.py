"""Reset bank (tg_config.reset_bank): the auto-reset of edge_follow / surface_follow takes a post-reset state computed ahead
of time on a second stream.  Whatever the bank does - off, always ready ("sync"), ready or not as the two streams happen to run ("on"),
never ready in time (one-step episodes) - every observation, terminal observation, reward, done flag, reset tick count and joint state must be
byte-identical: the same reset code runs either way, from the same RNG state (reference: edge_follow_env.py:311-336, base_surface_env.py:616-662,
robot.py:114-125, 188-260)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EDGE = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile",
            reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
SURF = dict(movement_mode="xyzRxRy", control_mode="TCP_velocity_control", noise_mode="simplex", observation_mode="tactile",
            reward_mode="dense", arm_type="ur5", tactile_sensor_name="digit")
VERT = dict(movement_mode="xRz", control_mode="TCP_velocity_control", noise_mode="vertical_simplex", observation_mode="tactile",
            reward_mode="dense", arm_type="mg400", tactile_sensor_name="tactip")
EDGE_POS = dict(EDGE, control_mode="TCP_position_control")


def rollout(env_id, modes, bank, n, max_steps, steps, act_dim, size=128):
    import tactile_gym_amd as tg
    venv = tg.make_vec(env_id, num_envs=n, max_steps=max_steps, image_size=[size, size], env_modes=modes, seed=11, auto_reset=True, reset_bank=bank)
    rng = np.random.default_rng(5)
    out = {"obs": [venv.reset()["tactile"].copy()], "rew": [], "done": [], "term": [], "ticks": [], "q": []}
    for _ in range(steps):
        a = rng.uniform(-0.25, 0.25, size=(n, act_dim)).astype(np.float32)
        obs, rew, done, infos = venv.step(a)
        out["obs"].append(obs["tactile"].copy()); out["rew"].append(rew.copy()); out["done"].append(done.copy())
        out["term"].append([np.asarray(infos[i]["terminal_observation"]["tactile"]).copy() if done[i] else None for i in range(n)])
        st = venv.get_state()
        out["ticks"].append(st["reset_ticks"].copy()); out["q"].append(st["q"].copy())
    stats = venv.bank_stats()
    venv.close()
    return out, stats


def same(a, b):
    for k in ("obs", "rew", "done", "ticks", "q"):
        for t, (x, y) in enumerate(zip(a[k], b[k])):
            assert np.array_equal(x, y), (k, t)
    for t, (x, y) in enumerate(zip(a["term"], b["term"])):
        for i, (u, v) in enumerate(zip(x, y)):
            assert (u is None) == (v is None) and (u is None or np.array_equal(u, v)), ("terminal observation", t, i)


@pytest.mark.parametrize("env_id,modes,act_dim,max_steps,steps", [
    ("edge_follow-v0", EDGE, 2, 7, 40),
    ("edge_follow-v0", EDGE_POS, 2, 5, 16),
    ("surface_follow-v0", SURF, 3, 6, 30),
    ("surface_follow-v2", VERT, 2, 5, 12),
])
def test_bank_off_sync_on_are_byte_identical(env_id, modes, act_dim, max_steps, steps):
    n = 48
    off, s_off = rollout(env_id, modes, "off", n, max_steps, steps, act_dim)
    syn, s_syn = rollout(env_id, modes, "sync", n, max_steps, steps, act_dim)
    aut, s_aut = rollout(env_id, modes, "on", n, max_steps, steps, act_dim)
    resets = int(sum(d.sum() for d in off["done"]))
    assert resets >= n * (steps // max_steps)
    assert s_off == {"mode": "off", "swapped": 0, "late": 0}
    # sync: the refill is waited for after every step, so only an env that finishes on the very step after its reset's refill ... never: all swapped
    assert s_syn["mode"] == "sync" and s_syn["swapped"] + s_syn["late"] == resets and s_syn["late"] == 0, s_syn
    assert s_aut["mode"] == "on" and s_aut["swapped"] + s_aut["late"] == resets, s_aut
    same(off, syn)
    same(off, aut)


def test_bank_not_ready_falls_back_to_the_reset_on_the_spot():
    """max_steps = 1: every env finishes on every step, the refill (every 8th step, on a stream of lower priority) cannot keep up: most resets are
    'late' - the join the step takes when an env finishes before its entry is ready - and nothing changes."""
    n = 32
    off, _ = rollout("edge_follow-v0", EDGE, "off", n, 1, 24, 2)
    aut, s = rollout("edge_follow-v0", EDGE, "on", n, 1, 24, 2)
    assert s["swapped"] + s["late"] == n * 24 and s["late"] > 0, s
    same(off, aut)


@pytest.mark.parametrize("object_mode", ["pole", "ball_on_plate"])
def test_object_balance_reset_template_equals_recomputed_reset(object_mode):
    """object_balance: Robot.reset drives the arm from the rest pose to a constant target under position motors that prescribe its velocity, and
    the object is teleported afterwards - the arm's post-reset state is the same for every reset, so k_reset_body computes it once
    (State.reset_tmpl).  Against reset_bank="off" (every reset recomputed with the fallen object still tied to the TCP) over 64 envs x 150
    steps with auto-resets: dones, reset tick counts and rewards identical, joints within 1e-13 rad (the last-bit residue of the full solver
    ticks of the recomputed resets), every frame within 3 pixels."""
    import tactile_gym_amd as tg
    modes = dict(movement_mode="xy", control_mode="TCP_velocity_control", object_mode=object_mode, rand_gravity=True, rand_embed_dist=True,
                 observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
    n, steps = 64, 150
    actions = np.random.default_rng(23).uniform(-0.25, 0.25, size=(steps, n, 2)).astype(np.float32)
    runs = []
    for bank in ("auto", "off"):
        v = tg.make_vec("object_balance-v0", num_envs=n, max_steps=60, image_size=[128, 128], env_modes=modes, seed=4100, auto_reset=True, reset_bank=bank)
        obs = v.reset()
        rec = dict(img=[obs["tactile"].copy()], q=[v.get_state()["q"].copy()], ticks=[v.get_state()["reset_ticks"].copy()], done=[], rew=[])
        for s in range(steps):
            obs, rew, done, _ = v.step(actions[s])
            st = v.get_state()
            rec["img"].append(obs["tactile"].copy()), rec["q"].append(st["q"].copy()), rec["ticks"].append(st["reset_ticks"].copy())
            rec["done"].append(done.copy()), rec["rew"].append(rew.copy())
        v.close()
        runs.append({k: np.asarray(x) for k, x in rec.items()})
    a, b = runs
    assert np.array_equal(a["done"], b["done"]) and a["done"].sum() >= n
    assert np.array_equal(a["ticks"], b["ticks"]) and np.array_equal(a["rew"], b["rew"])
    assert np.abs(a["q"] - b["q"]).max() < 1e-13, np.abs(a["q"] - b["q"]).max()
    assert (a["img"] != b["img"]).reshape(steps + 1, n, -1).sum(-1).max() <= 3


def test_auto_is_on_and_a_reset_env_keeps_its_wavefront_on_the_light_path():
    """Round 5.  reset_bank="auto" is the bank for the UR5 as well, and a reset renews the solver licence instead
    of dropping it.  What can be seen from outside: (a) bank_stats says "on"; (b) resetting ONE env of a wavefront leaves the other 63 envs'
    trajectories where they were - they share that env's licence decision (it is the wavefront's), and whether a tick takes the analytic fixed
    point or the full solve may move a joint by the solve's last bits only (1e-11 rad is the bound the default-vs-literal test uses), never by
    more; (c) the reset env itself follows the same trajectory as an env of a batch that was reset as a whole (same seed, same RNG stream)."""
    import tactile_gym_amd as tg
    n = 64

    def run(reset_one):
        v = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=EDGE, seed=3, auto_reset=True)
        v.reset()
        rng = np.random.default_rng(9)
        qs = []
        for k in range(30):
            if reset_one and k == 10:
                m = np.zeros(n, np.uint8); m[5] = 1
                v.reset(m)
            v.step(rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32))
            qs.append(v.get_state()["q"].copy())
        mode = v.bank_stats()["mode"]
        v.close()
        return np.stack(qs), mode
    a, mode = run(False)
    b, _ = run(True)
    assert mode == "on"
    qa, qb = (a, b) if a.shape[-1] == n else (np.swapaxes(a, -1, -2), np.swapaxes(b, -1, -2))      # [..., joints, env]
    others = [e for e in range(n) if e != 5]
    assert np.max(np.abs(qa[..., others] - qb[..., others])) < 1e-11
    assert np.max(np.abs(qa[10:, :, 5] - qb[10:, :, 5])) > 1e-4          # env 5 did restart from its reset pose


@pytest.mark.parametrize("env_id,modes,act_dim", [("edge_follow-v0", EDGE, 2), ("surface_follow-v0", SURF, 3)])
def test_bank_modes_are_byte_identical_with_the_episodes_out_of_phase(env_id, modes, act_dim):
    """The condition an RL run is in and test_bank_off_sync_on_are_byte_identical is not: the envs' episodes end in DIFFERENT steps (masked resets
    put three groups of envs out of phase), so in nearly every step some envs of a wavefront swap in a bank entry - with its solver licence and
    exact sines / cosines (round 5) - while their neighbours are mid-episode, and the refill of spent entries runs beside later steps.  Bank off /
    sync / on (entries ready or not as the two streams happen to run): every observation, reward, done flag, reset tick count and joint state
    byte-identical."""
    import tactile_gym_amd as tg
    n, max_steps, steps = 96, 11, 70

    def run(bank):
        venv = tg.make_vec(env_id, num_envs=n, max_steps=max_steps, image_size=[128, 128], env_modes=modes, seed=21, auto_reset=True, reset_bank=bank)
        rng = np.random.default_rng(8)
        out = {"obs": [venv.reset()["tactile"].copy()], "rew": [], "done": [], "ticks": [], "q": []}
        for k in range(steps):
            if k in (2, 5, 7):                                   # envs 0, 3, 6, .. / 1, 4, .. / every 5th restart here: three phases + the rest
                m = np.zeros(n, np.uint8)
                m[{2: slice(0, n, 3), 5: slice(1, n, 3), 7: slice(0, n, 5)}[k]] = 1
                venv.reset(m)
            obs, rew, done, _ = venv.step(rng.uniform(-0.25, 0.25, size=(n, act_dim)).astype(np.float32))
            st = venv.get_state()
            out["obs"].append(obs["tactile"].copy()); out["rew"].append(rew.copy()); out["done"].append(done.copy())
            out["ticks"].append(st["reset_ticks"].copy()); out["q"].append(st["q"].copy())
        stats = venv.bank_stats()
        venv.close()
        return out, stats
    off, _ = run("off")
    syn, s_syn = run("sync")
    aut, s_aut = run("auto")
    per_step = [int(d.sum()) for d in off["done"][12:]]
    assert sum(1 for c in per_step if 0 < c < n) >= len(per_step) // 3 and max(per_step) < n     # four phases over 11-step episodes: some envs finish in a third of the steps, never all
    assert s_syn["late"] == 0 and s_syn["swapped"] > n and s_aut["mode"] == "on" and s_aut["swapped"] + s_aut["late"] == s_syn["swapped"]
    for other in (syn, aut):
        for key in ("obs", "rew", "done", "ticks", "q"):
            for t, (x, y) in enumerate(zip(off[key], other[key])):
                assert np.array_equal(x, y), (key, t)


def test_object_push_reset_template_equals_the_literal_reset(monkeypatch):
    """Round 6: object_push's Robot.reset (robots/arms/robot.py:114-125, 188-260) with the previous episode's cube clear of the tip runs arm-only
    ticks with the cube frozen ("optimistic" move) and, once one such move has run through, takes its result as a template without any tick
    (csrc/tg_contact_wave.hip: k_reset_contact_wave).  Against the literal move (TG_LITERAL_RESET=1: every tick a full contact tick) and against
    the optimistic move without the template (TG_RESET_BANK=0) over rollouts with many auto-resets - short episodes, so that resets happen both
    with the cube still at the tip (the test fails and the move is literal) and with it pushed away: dones, reset tick counts, contact counts
    exact; joints to rounding (the arm-only tick is another order of the same sums); images at most one grey level off on a few pixels."""
    import tactile_gym_amd as tg
    from test_gpu_config_scale import PUSH
    n, steps = 128, 90
    acts = np.random.default_rng(2).uniform(-0.25, 0.25, size=(steps, n, 2)).astype(np.float32)

    def rollout(env):
        for k in ("TG_LITERAL_RESET", "TG_RESET_BANK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        v = tg.make_vec("object_push-v0", num_envs=n, max_steps=25, image_size=[128, 128], env_modes=PUSH, seed=77, auto_reset=True)
        v.reset()
        out = dict(q=[], done=[], ticks=[], img=[], cc=[])
        for s in range(steps):
            obs, rew, done, info = v.step(acts[s])
            st = v.get_state()
            out["q"].append(st["q"].copy()), out["done"].append(done.copy()), out["ticks"].append(st["reset_ticks"].copy())
            out["img"].append(obs["tactile"][..., 0].copy()), out["cc"].append(st["contact_count"].copy())
        stats = v.bank_stats()
        v.close()
        return {**{k: np.asarray(x) for k, x in out.items()}, "stats": stats}
    tmpl, opt, lit = rollout({}), rollout({"TG_RESET_BANK": "0"}), rollout({"TG_LITERAL_RESET": "1"})
    assert tmpl["stats"]["mode"] == "template" and tmpl["stats"]["swapped"] > n and tmpl["stats"]["late"] >= n, tmpl["stats"]   # both routes were taken
    assert opt["stats"]["swapped"] == 0 and lit["stats"]["swapped"] == 0
    assert lit["done"].sum() >= 3 * n                                      # every env was reset several times
    assert len(set(lit["ticks"].ravel().tolist())) >= 1
    for other, name in ((tmpl, "template"), (opt, "optimistic")):
        assert np.array_equal(other["done"], lit["done"]) and np.array_equal(other["ticks"], lit["ticks"]), name
        assert np.array_equal(other["cc"], lit["cc"]), name
        assert np.abs(other["q"] - lit["q"]).max() < 1e-10, (name, np.abs(other["q"] - lit["q"]).max())
        d = other["img"].astype(np.int16) - lit["img"].astype(np.int16)
        assert np.abs(d).max() <= 1 and (d != 0).reshape(steps * n, -1).sum(1).max() <= 16, name
    assert np.array_equal(tmpl["q"], opt["q"])                             # the template IS the optimistic move's result, bit for bit
