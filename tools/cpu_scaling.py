"""How the CPU oracle scales with the number of pinned processes on this box (what bounds bench.py's cpu_baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
import bench

def run(k, seconds=4.0):
    cpus = sorted(os.sched_getaffinity(0))
    cpus = cpus[::max(1, len(cpus) // k)][:k]
    ctx = mp.get_context("fork")
    barrier, q = ctx.Barrier(k), ctx.Queue()
    ps = [ctx.Process(target=bench._cpu_worker, args=(i, cpus[i], seconds, 1, barrier, q), daemon=True) for i in range(k)]
    [p.start() for p in ps]
    res = [q.get() for _ in range(k)]
    [p.join() for p in ps]
    tot = sum(o[0][0] for _, o in res) / max(o[0][1] for _, o in res)
    return tot

if __name__ == "__main__":
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            print(f, open(f).read().strip())
        except OSError as e:
            print(f, "-", e.strerror)
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "quota", bench._cgroup_cpu_quota())
    for k in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if k <= len(os.sched_getaffinity(0)):
            t = run(k)
            print(f"{k:4d} processes: {t:9.1f} env-steps/s total, {t / k:8.1f} per process", flush=True)
