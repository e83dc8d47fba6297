import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle.ref_env import OracleEdgeFollowEnv
from oracle import pb_math as pm
from tactile_gym_amd import hip_ops
from tactile_gym_amd.rl_envs.edge_follow import REST_POSES
from tactile_gym_amd.robot_model import load_tgmodel, make_robot
tg=load_tgmodel('ur5','standard','tactip'); rest=np.array(REST_POSES['ur5']['tactip']['standard']); robot=make_robot(tg,rest,'tactip')
env=OracleEdgeFollowEnv(seed=1)
n=16; rng=np.random.default_rng(0)
tps=[];trs=[];refs=[]
for i in range(n):
    embed=rng.uniform(0.0015,0.0065)
    tpos,trpy=env._work_to_world(np.array([0,0,embed]),np.zeros(3)); torn=pm.quat_from_euler(trpy)
    env.arm.reset_joint_states(rest)
    refs.append(env.arm.inverse_kinematics('tcp_link',tpos,torn,100,1e-8)); tps.append(tpos); trs.append(pm.mat_from_quat(torn))
q,it=hip_ops.inverse_kinematics(robot,np.tile(rest,(n,1)),np.array(tps),np.array(trs))
print('iters',it); print('dq',np.abs(q-np.array(refs)).max(1))
# residual check
for i in range(3):
    p,_,_,_,R=env.arm.link_state('tcp_link',q=q[i],qd=np.zeros(6)); print('gpu sol resid',np.abs(p-tps[i]).max(), np.abs(R-trs[i]).max())
    p,_,_,_,R=env.arm.link_state('tcp_link',q=refs[i],qd=np.zeros(6)); print('ora sol resid',np.abs(p-tps[i]).max(), np.abs(R-trs[i]).max())
