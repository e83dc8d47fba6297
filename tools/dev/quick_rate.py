"""Development aid: rate of one configuration (device-resident random rollout), optionally in threshold mode.  python tools/dev/quick_rate.py env size n steps [thr] [k=v ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench


def main():
    env_id, size, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    thr = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    kw = dict(a.split("=") for a in sys.argv[6:])
    import torch

    def barrier():
        torch.cuda.synchronize()
    out = bench.companion(env_id, size, n, "f64", steps, barrier, "", residual_threshold=thr, **kw)
    print(json.dumps({k: out[k] for k in out if k != "roofline"} | {"kernel_ms": out["roofline"]["kernel_ms"]}))


if __name__ == "__main__":
    main()
