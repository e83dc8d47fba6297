import numpy as np, sys
sys.path.insert(0,'/root/repo')
import tactile_gym_amd as tg
from oracle.ref_env import OracleEdgeFollowEnv
modes = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
n=8
venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128,128], env_modes=modes, seed=11, auto_reset=False)
oracles=[OracleEdgeFollowEnv(seed=11+i, max_steps=200, image_size=(128,128), env_modes=modes) for i in range(n)]
obs=venv.reset(); ref=[o.reset() for o in oracles]
st=venv.get_state()
np.set_printoptions(precision=3, linewidth=200)
for i,o in enumerate(oracles):
    print(i,'ang eq',st['edge_ang'][i]==o.edge_ang,'embed eq',st['embed_dist'][i]==o.embed_dist,'ticks',st['reset_ticks'][i],o.reset_ticks,'dq',np.abs(st['q'][i]-o.arm.q).max(), 'dqd', np.abs(st['qd'][i]-o.arm.qd).max(),'px',(obs['tactile'][i]!=ref[i]['tactile']).sum())
rng=np.random.default_rng(12)
for step in range(4):
    a=rng.uniform(-0.25,0.25,size=(n,2)).astype(np.float32)
    obs,rew,done,info=venv.step(a); st=venv.get_state()
    for i,o in enumerate(oracles):
        ro,rr,rd,_=o.step(a[i])
        print(step,i,'dqdtarget',np.abs(st['qd_target'][i]-o.last_req_joint_vels).max(),'dq',np.abs(st['q'][i]-o.arm.q).max(),'dqd',np.abs(st['qd'][i]-o.arm.qd).max(),'drew',abs(rew[i]-rr),'px',(obs['tactile'][i]!=ro['tactile']).sum())
