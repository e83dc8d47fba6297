"""The per-step exchange to rank 0 on the device: the tile-sparse payload kernels against their torch definition, and the
ipc transport (peers storing straight into rank 0's receive slots, stream-side flags) with two processes.  The GPU box has ONE GPU, so
the two ranks share it: what is exercised is the mechanism (IPC handles, slot layout, flags, the one-step-behind pipeline, every
payload), not the xGMI hop - RCCL itself refuses two ranks on one device, so the control plane of these tests is gloo."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EDGE = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile",
            reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
BAL = dict(movement_mode="xy", control_mode="TCP_velocity_control", object_mode="pole", rand_gravity=True, rand_embed_dist=True,
           observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
PUSH = dict(movement_mode="TyRz", control_mode="TCP_velocity_control", rand_init_orn=False, rand_obj_mass=False, traj_type="simplex",
            observation_mode="tactile_and_feature", reward_mode="dense", arm_type="mg400", tactile_sensor_name="digitac")


@pytest.mark.parametrize("env_id,modes,size,n", [("edge_follow-v0", EDGE, 128, 1024), ("edge_follow-v0", EDGE, 64, 37),
                                                ("object_balance-v0", BAL, 256, 130), ("object_push-v0", PUSH, 128, 64)])
def test_tile_kernels_match_definition(env_id, modes, size, n):
    """tg_pack_tiles == parallel.torch_pack_tiles as a SET of records (the kernel's record order is not defined), the header's count
    exact; tg_unpack_tiles restores every image bit for bit; the counters are left zero (a second launch gives the same message)."""
    import torch
    import tactile_gym_amd as tg
    from tactile_gym_amd.parallel import TILE_REC, TorchShard, torch_pack_tiles
    v = tg.make_vec(env_id, num_envs=n, max_steps=200, image_size=[size, size], env_modes=modes, seed=3, obs_mode="torch", auto_reset=True)
    sh = TorchShard(v)
    sh.reset()
    rng = np.random.default_rng(0)
    for _ in range(3):
        sh.step(torch.from_numpy(rng.uniform(-0.25, 0.25, size=(n, v.act_dim)).astype(np.float32)).cuda())
    tac, tmpl = v.tactile_torch(), sh.tile_template()
    T = (size // 16) ** 2
    cap = 16 + TILE_REC * n * T
    counters = torch.zeros(4, dtype=torch.int32, device="cuda")
    ref = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    count = torch_pack_tiles(torch, tac, tmpl, ref)
    assert 0 < count < n * T
    for rep in range(2):
        msg = torch.full((cap,), 0xAB, dtype=torch.uint8, device="cuda")
        sh.pack_tiles(msg.data_ptr(), counters)
        torch.cuda.synchronize()
        hdr = msg[:16].view(torch.int32).tolist()
        assert hdr == [count, n, T, 0x54475431], (hdr, count)
        assert counters.tolist() == [0, 0, 0, 0]
        got = msg[16:16 + TILE_REC * count].reshape(count, TILE_REC)
        want = ref[16:16 + TILE_REC * count].reshape(count, TILE_REC)
        order = torch.argsort(got[:, :4].contiguous().view(torch.int32).reshape(-1))
        assert torch.equal(got[order], want)                 # torch_pack_tiles emits ids in increasing order
        assert bool((msg[16 + TILE_REC * count:] == 0xAB).all())   # nothing written beyond the records
        out = torch.full((n, size * size), 0xCD, dtype=torch.uint8, device="cuda")
        sh.unpack_tiles(msg.data_ptr(), n, out.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(out.reshape(tac.shape), tac)
    frac = (16 + TILE_REC * count) / tac.numel()
    print(f"{env_id} {size}x{size}, {n} envs: {count} of {n * T} tiles live, tile message = {100 * frac:.1f} % of the full images")
    v.close()


def test_flag_wait_times_out_instead_of_hanging():
    """A wait on a flag nobody raises ends after its timeout and reports the lane in the error word; a raised flag passes at once."""
    import ctypes as C
    import time
    import torch
    from tactile_gym_amd import _capi as capi
    L = capi.lib()
    flags = torch.zeros(64, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(L.tg_flag_set(s, C.c_void_p(flags.data_ptr()), 2, 16, 5))
    capi.check(L.tg_flag_wait(s, C.c_void_p(flags.data_ptr()), 2, 16, 5, C.c_void_p(err.data_ptr()), 2000))
    torch.cuda.synchronize()
    assert flags[0].item() == 5 and flags[16].item() == 5 and err.item() == 0
    t0 = time.perf_counter()
    capi.check(L.tg_flag_wait(s, C.c_void_p(flags.data_ptr()), 3, 16, 6, C.c_void_p(err.data_ptr()), 300))    # lanes 0, 1 hold 5; lane 2 holds 0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert 0.25 < dt < 3.0, dt
    assert err.item() == 0b111
    # once the error word is set, later waits on this stream return at once: a stuck partner costs ONE timeout, not one per queued wait
    t0 = time.perf_counter()
    for _ in range(20):
        capi.check(L.tg_flag_wait(s, C.c_void_p(flags.data_ptr()), 3, 16, 7, C.c_void_p(err.data_ptr()), 300))
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.25
    assert err.item() == 0b111


def _ipc_worker(rank, world, port, out_path, env_id, modes, size, n_local, payload, overlap, steps):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import tactile_gym_amd as tg
    from tactile_gym_amd.parallel import ShardedVecEnv, TorchShard
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v = tg.make_vec(env_id, num_envs=n_local, max_steps=200, image_size=[size, size], env_modes=modes, seed=50 + rank * n_local,
                    obs_mode="torch", auto_reset=True)
    shard = TorchShard(v, pipelined=True)
    env = ShardedVecEnv(shard, dist, overlap=overlap, payload=payload, transport="ipc", timeout_ms=20000)
    gen = torch.Generator().manual_seed(5)
    acts = [((torch.rand(world * n_local, v.act_dim, generator=gen) - 0.5) * 0.5) for _ in range(steps)]
    hist = []
    with torch.cuda.stream(shard.stream):
        obs = env.reset()
        assert env.transport == "ipc" and env.payload == payload
        hist.append({k: t.clone() for k, t in obs.items()})
        for k in range(steps):
            obs, rew, done, _ = env.step(acts[k][rank * n_local:(rank + 1) * n_local].cuda())
            hist.append(dict({kk: t.clone() for kk, t in obs.items()}, rew=rew.clone(), done=done.clone()))
        if overlap:
            last = env.flush()
            if rank == 0:
                hist.append(dict({kk: t.clone() for kk, t in last[0].items()}, rew=last[1].clone(), done=last[2].clone()))
        torch.cuda.synchronize()
    info = env.exchange_info()
    if rank == 0:
        # round 5: with the tile payload rank 0's shard is drawn straight into its block of the gathered batch (tg_set_obs_targets: two render
        # targets, one per alternating batch) - the comparison below is against a context that draws into its own buffer
        assert info["rank0_draws_into_batch"] == (payload == "tiles")
        # the single-process reference: the same world * n_local envs (seeds 50 ...) in one context, same actions
        ref = tg.make_vec(env_id, num_envs=world * n_local, max_steps=200, image_size=[size, size], env_modes=modes, seed=50, obs_mode="torch",
                          auto_reset=True)
        rs = TorchShard(ref)
        want = [{k: t.clone() for k, t in rs.reset().items()}]
        for k in range(steps):
            o, r, d, _ = rs.step(acts[k].cuda())
            want.append(dict({kk: t.clone() for kk, t in o.items()}, rew=r.clone(), done=d.clone()))
        # overlap: step k hands out the batch of step k - 1 (the first step: the local shard's own view), flush() the last one
        got = [hist[0]] + (hist[2:] if overlap else hist[1:])
        exp = want
        assert len(got) == len(exp) and len(got) >= 3
        for a, b in zip(got, exp):
            assert set(a) == set(b)
            for key in b:
                assert a[key].shape == b[key].shape and torch.equal(a[key], b[key]), key
        torch.save({"info": info, "frames": len(got)}, out_path)
        ref.close()
    dist.barrier()
    env.close()
    v.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("env_id,modes,size,payload,overlap", [("edge_follow-v0", EDGE, 128, "tiles", True), ("edge_follow-v0", EDGE, 128, "interior", True),
                                                               ("edge_follow-v0", EDGE, 128, "full", False), ("object_push-v0", PUSH, 128, "tiles", True),
                                                               ("edge_follow-v0", dict(EDGE, observation_mode="visuotactile"), 128, "tiles", True)])
def test_ipc_exchange_two_ranks(tmp_path, env_id, modes, size, payload, overlap):
    """Two ranks (two processes, both on this box's one GPU), ipc transport: every batch rank 0 is handed equals the single-process batch of
    the same 2 x n envs bit for bit - tactile images, reward, done, extended_feature (object_push), visual (visuotactile)."""
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_ipc_worker, args=(2, port, out, env_id, modes, size, 48, payload, overlap, 6), nprocs=2, join=True)
    got = torch.load(out)
    assert got["info"]["transport"] == "ipc" and got["info"]["payload"] == payload and got["frames"] >= 6
    if payload == "tiles":
        assert len(got["info"]["message_bytes_last"]) == 2 and max(got["info"]["message_bytes_last"]) < got["info"]["message_bytes_capacity"]
    print(got["info"])


def test_unpack_multi_restores_only_what_the_last_message_touched():
    """tg_unpack_tiles_multi with a previous-ids list: three successive messages of two "ranks" (two halves of one 256-env batch, moving
    contact patches, an auto-reset in between) into the same destination - after each, the destination equals the images exactly; the
    skipped rank's block is never written; the tail rides in the pack launch."""
    import torch
    import tactile_gym_amd as tg
    from tactile_gym_amd.parallel import TILE_REC, TorchShard
    n, size = 128, 128
    envs = [tg.make_vec("edge_follow-v0", num_envs=n, max_steps=3, image_size=[size, size], env_modes=EDGE, seed=30 + 1000 * r, obs_mode="torch", auto_reset=True)
            for r in range(2)]
    shards = [TorchShard(v) for v in envs]
    T = (size // 16) ** 2
    stride = ((16 + TILE_REC * n * T + 15) // 16) * 16 + 4096
    msgs = torch.zeros(3 * stride, dtype=torch.uint8, device="cuda")          # rank 0 (skipped), ranks 1 and 2
    dst = shards[0].tile_template().reshape(1, -1).repeat(3 * n, 1).contiguous()
    dst[:n] = 0x5A                                                            # the skipped rank's block: must stay as it is
    prev = torch.zeros(3 * (n * T + 1), dtype=torch.int32, device="cuda")
    counters = torch.zeros(4, dtype=torch.int32, device="cuda")
    tail = torch.arange(4096 - 7, dtype=torch.int32, device="cuda").to(torch.uint8)
    rng = np.random.default_rng(2)
    for sh in shards:
        sh.reset()
    for k in range(5):
        for r, sh in enumerate(shards):
            sh.step(torch.from_numpy(rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)).cuda())
            sh.pack_tiles(msgs[(r + 1) * stride:].data_ptr(), counters, tail=tail, tail_offset=stride - 4096)
        shards[0].unpack_tiles_multi(msgs.data_ptr(), stride, 3, 0, n, dst.data_ptr(), prev.data_ptr())
        torch.cuda.synchronize()
        for r, v in enumerate(envs):
            assert torch.equal(dst[(r + 1) * n:(r + 2) * n].reshape(v.tactile_torch().shape), v.tactile_torch()), (k, r)
            assert torch.equal(msgs[(r + 1) * stride + stride - 4096:(r + 1) * stride + stride - 7], tail)
        assert bool((dst[:n] == 0x5A).all())
        counts = prev.reshape(3, -1)[:, 0].tolist()
        assert counts[0] == 0 and 0 < counts[1] < n * T // 2 and 0 < counts[2] < n * T // 2
    for v in envs:
        v.close()


def _ipc_stuck_worker(rank, world, port, out_path, overlap):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import tactile_gym_amd as tg
    from tactile_gym_amd.parallel import ShardedVecEnv, TorchShard
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local = 16
    v = tg.make_vec("edge_follow-v0", num_envs=n_local, max_steps=200, image_size=[128, 128], env_modes=EDGE, seed=70 + rank * n_local, obs_mode="torch", auto_reset=True)
    shard = TorchShard(v, pipelined=True)
    env = ShardedVecEnv(shard, dist, overlap=overlap, payload="tiles", transport="ipc", timeout_ms=300)
    a = torch.zeros(n_local, v.act_dim, device="cuda")
    raised_at, msg = None, ""
    with torch.cuda.stream(shard.stream):
        env.reset()
        for k in range(40):
            if rank == 1 and k >= 2:
                break                                  # the peer stops sending: rank 0's waits on its ready flag time out
            try:
                env.step(a)
            except RuntimeError as e:
                raised_at, msg = k, str(e)
                break
        torch.cuda.synchronize()
    dist.barrier()
    try:
        env.close()
    except RuntimeError as e:
        if raised_at is None:
            raised_at, msg = "close", str(e)
    if rank == 0:
        torch.save({"raised_at": raised_at, "msg": msg}, out_path)
    v.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_a_stuck_peer_raises_on_the_step_path(tmp_path, overlap):
    """a peer that stops sending.  overlap=False: the step that would have handed out a stale slot raises; overlap=True: the
    error word is polled every 16th step without a synchronisation (the host runs ahead of the device there, so a short run may end before a
    poll has seen it) and close() raises whatever no step has reported - never silent."""
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_ipc_stuck_worker, args=(2, port, out, overlap), nprocs=2, join=True)
    got = torch.load(out)
    assert got["raised_at"] is not None and "timed out" in got["msg"], got
    assert got["raised_at"] == 2 if not overlap else (got["raised_at"] == "close" or 2 <= got["raised_at"] <= 39), got
