import numpy as np, sys
sys.path.insert(0, "/root/repo")
import tactile_gym_amd as tg, bench
v = tg.make_vec("object_balance-v0", num_envs=1024, max_steps=250, image_size=[256,256], env_modes=bench.BAL_MODES, seed=1, auto_reset=True)
v.reset()
rng = np.random.default_rng(0)
ticks=[]; dones=[]
for s in range(120):
    o,r,d,i = v.step(rng.uniform(-0.25,0.25,(1024,2)).astype(np.float32))
    dones.append(d.sum())
    if d.any(): ticks += list(v.get_state()["reset_ticks"][d])
ticks=np.array(ticks)
print("dones per step mean", np.mean(dones), "steps with a done", np.mean(np.array(dones)>0), "reset ticks: mean", ticks.mean(), "min", ticks.min(), "max", ticks.max(), "pcts", np.percentile(ticks,[10,50,90]))
