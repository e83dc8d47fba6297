"""Host-side cost of one pipelined step (edge_follow, N envs): the enqueue loop timed WITHOUT waiting for the device, then the device drain.
    python tools/host_cost.py [num_envs] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
w = bench.Workload("edge_follow-v0", n, 128, "f64", 0, 0)
with w.on_stream():
    w.shard.reset()
    for _ in range(100):
        w.shard.step(w.actions())
    torch.cuda.synchronize()
    for what in ("step+sample", "sample only", "step only", "step_async only"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if what == "step+sample":
            for _ in range(K):
                w.shard.step(w.actions())
        elif what == "sample only":
            for _ in range(K):
                w.actions()
        elif what == "step only":
            a = w.act_buf
            for _ in range(K):
                w.shard.step(a)
        else:
            a = w.act_buf
            for _ in range(K):
                w.venv.step_async(a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{what:16s} n={n}: host enqueue {1e6*(t1-t0)/K:7.2f} us/step, drained after {1e6*(t2-t0)/K:7.2f} us/step")
w.close()
