"""Development: with a library built with -DTG_TL_STAMPS, prints the mean launch-start intervals of the last 500 pipelined steps at close."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = sys.argv[2] if len(sys.argv) > 2 else "edge_follow-v0"
w = bench.Workload(env, n, 256 if env == "object_balance-v0" else 128, "f64", 0, 0)
with w.on_stream():
    w.shard.reset()
    for _ in range(310):
        w.shard.step(w.actions())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        w.shard.step(w.actions())
    torch.cuda.synchronize()
    print(f"{env} n={n}: {1e3 * (time.perf_counter() - t0):.3f} us/step", file=sys.stderr)
w.close()
