import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_pool import oracle_rollouts
from test_gpu_config_scale import BAL, _hip_rollout
def main():
    thr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-7
    n, steps, seed = 64, 60, 5200
    actions = np.random.default_rng(13).uniform(-0.25, 0.25, size=(120, n, 2)).astype(np.float32)[:steps]
    extra = dict(solver_residual_threshold=thr) if thr else {}
    hip = _hip_rollout("object_balance-v0", BAL, 128, 250, n, seed, actions, auto_reset=True, **extra)
    ref = oracle_rollouts("OracleObjectBalanceEnv", dict(max_steps=250, image_size=(128, 128), env_modes=BAL, **extra), seed, actions, auto_reset=True)
    for i, r in enumerate(ref):
        d = (hip["img"][:, i] != r["img"]).reshape(steps + 1, -1).sum(1)
        if d.any():
            f = np.nonzero(d)[0]
            print("env", i, "frames", f, "pixels", d[f], "dones", np.nonzero(r["done"])[0], "maxabs", np.abs(hip["img"][f, i].astype(int) - r["img"][f].astype(int)).max())
            for ff in f:
                print("  frame", ff, "dq", np.abs(hip["q"][ff, i] - r["q"][ff]).max(), "dbody", np.abs(hip["body"][ff - 1, i] - r["body"][ff - 1]).max() if ff > 0 else None,
                      "dxf", np.abs(hip["xf"][ff, i].astype(np.float64) - r["xf"][ff].astype(np.float64)).max())
    print("done")


if __name__ == "__main__":
    main()
